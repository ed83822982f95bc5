"""Mirror of the reference's `model` package (spherical_model.py / spherical_model_iterative.py)."""
