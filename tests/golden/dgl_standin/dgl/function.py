"""dgl.function builtins used by the reference hot path (descriptors only)."""


class _Msg:
    def __init__(self, kind, a, b, out):
        self.kind, self.a, self.b, self.out = kind, a, b, out


class _Red:
    def __init__(self, kind, msg, out):
        self.kind, self.msg, self.out = kind, msg, out


def u_add_v(a, b, out):
    return _Msg("u_add_v", a, b, out)


def u_mul_e(a, b, out):
    return _Msg("u_mul_e", a, b, out)


def copy_e(a, out):
    return _Msg("copy_e", a, None, out)


def sum(msg, out):  # noqa: A001 (DGL's own name)
    return _Red("sum", msg, out)
