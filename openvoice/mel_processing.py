from openvoice_amd.mel_processing import spectrogram_torch  # noqa: F401
