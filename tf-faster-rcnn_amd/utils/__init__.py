"""Function mirrors of the reference's ``utils/`` package (hot-path functions only)."""
