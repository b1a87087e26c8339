"""Empty stand-in (imported by uf3.data.io only)."""
