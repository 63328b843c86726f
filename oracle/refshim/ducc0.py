"""Stand-in for `ducc0` (spherical-harmonic transforms; off the hot path, never called)."""
