"""Process-wide switches of the MI355X op library."""

ARITH_STRICT, ARITH_FMA, ARITH_FMA_ALT = 0, 1, 2

# Squared-distance contraction hypothesis used by FPS and ball query (see
# include/pn2_abi.h and DESIGN.md "Arithmetic modes").  ARITH_FMA reproduces what
# nvcc's default --fmad=true does to the reference expression.
arith_mode = ARITH_FMA
