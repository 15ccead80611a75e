from openvoice_amd.utils import HParams, get_hparams_from_file  # noqa: F401
