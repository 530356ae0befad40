"""Build-container-only harness that imports the UNMODIFIED reference from /root/reference and
dumps golden vectors.  Never imported by tests/, bench.py, smoke() or the product."""
