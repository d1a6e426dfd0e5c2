"""Drop-in for the `simple_knn` package (MANUS: src/models/gaussian.py:4)."""
