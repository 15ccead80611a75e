from openvoice_amd.api import BaseSpeakerTTS, OpenVoiceBaseClass, ToneColorConverter  # noqa: F401
