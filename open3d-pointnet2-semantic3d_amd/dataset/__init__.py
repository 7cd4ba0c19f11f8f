"""Device-side scene sampling (mirror of the reference's dataset/ package for the step before the SA/FP stack)."""
from . import semantic_dataset  # noqa: F401
from .semantic_dataset import SemanticFileData  # noqa: F401
