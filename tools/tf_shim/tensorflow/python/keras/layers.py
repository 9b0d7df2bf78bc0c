from tensorflow import (Add, Conv2D, Cropping2D, Dense, ELU, InputSpec, KLayer as Layer, Lambda, MaxPool2D, Multiply,  # noqa: F401
                        TimeDistributed, UpSampling2D)
