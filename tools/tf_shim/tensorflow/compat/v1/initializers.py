def orthogonal(*a, **kw):
    return ("orthogonal",)
