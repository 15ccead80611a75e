from openvoice_amd.se_extractor import get_se, hash_numpy_array  # noqa: F401
