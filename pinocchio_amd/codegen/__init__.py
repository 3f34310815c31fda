"""Model compiler: turns one constant model into straight-line HIP kernel bodies (RNEA, ABA,
FK + Jacobians) with every structural zero folded away.  See emit.py."""
