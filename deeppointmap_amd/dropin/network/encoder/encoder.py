"""reference import path network.encoder.encoder -> MI355X Encoder (same ctor, state dict, call contract)."""
from deeppointmap_amd.encoder import Encoder  # noqa: F401
