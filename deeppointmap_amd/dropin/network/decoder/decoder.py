"""reference import path network.decoder.decoder -> MI355X Decoder."""
from deeppointmap_amd.decoder import Decoder  # noqa: F401
