"""TEST INFRASTRUCTURE ONLY — message/reduce descriptors for the stand-in dgl."""


def copy_src(src, out):
    return ('copy_u', src, out)


copy_u = copy_src


def u_mul_e(lhs, rhs, out):
    return ('u_mul_e', lhs, rhs, out)


def sum(msg, out):  # noqa: A001 - mirrors dgl.function.sum
    return ('sum', msg, out)
