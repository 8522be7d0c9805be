from sage355.encoders import Encoder  # noqa: F401
