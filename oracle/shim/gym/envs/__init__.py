from . import registration
from .registration import make, register, registry
