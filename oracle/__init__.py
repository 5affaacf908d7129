"""CPU oracle package -- TEST INFRASTRUCTURE ONLY (see mlt_oracle.c header)."""
from .oracle import Oracle, build  # noqa: F401
