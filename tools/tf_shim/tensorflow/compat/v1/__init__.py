from tensorflow import *  # noqa: F401,F403
