from .glfsr import GLFSR
from .pn import GALOIS_LFSR_POLYS, PNSequence, generate_mask

__all__ = ["GLFSR", "PNSequence", "GALOIS_LFSR_POLYS", "generate_mask"]
