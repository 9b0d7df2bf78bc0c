def binary_crossentropy(*a, **kw):
    raise NotImplementedError
