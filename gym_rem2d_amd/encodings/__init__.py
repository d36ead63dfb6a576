"""Genotype -> phenotype encodings (host side input generators for the batched stepper)."""
from .direct import DirectEncoding  # noqa: F401
from .lsystem import LSystem  # noqa: F401
from .network import NNEncoding, FeedForwardCPPN  # noqa: F401
