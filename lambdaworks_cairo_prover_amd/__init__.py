"""MI355X-native STARK proving hot path for the lambdaworks Cairo prover (Stark252 NTT/LDE, Keccak Merkle
commitments, constraint composition, DEEP/FRI) behind the C ABI of include/stark252_hip.h."""
from . import _lib  # noqa: F401
from .api import *  # noqa: F401,F403
