from tensorflow import _KBackend

std, learning_phase, int_shape, mean = _KBackend.std, _KBackend.learning_phase, _KBackend.int_shape, _KBackend.mean
