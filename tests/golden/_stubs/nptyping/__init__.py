class _Sub(type):
    def __getitem__(cls, item):
        return cls


class NDArray(metaclass=_Sub):
    pass


class Shape(metaclass=_Sub):
    pass


Bool = Float32 = UInt32 = UInt64 = object
