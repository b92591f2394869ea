"""Drop-in shim package (see shim/README.md)."""
