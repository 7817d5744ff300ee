"""The reference's `training` package is CLI mains, data loading and logging (out of scope); only the
generic evaluation loop that sits directly on the hot path's outputs is mirrored here."""
