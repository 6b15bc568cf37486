"""Layer mirrors of the reference's ``models/`` package (hot-path layers only)."""
