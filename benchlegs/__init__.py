"""The legs of bench.py, one module per BASELINE configuration (bench.py parses the arguments, brings up the ranks and dispatches)."""
