from openvoice_amd.models import SynthesizerTrn  # noqa: F401
