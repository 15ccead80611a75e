"""``get_se`` -- tone-colour embedding of a reference recording (thin shim).

Keeps the reference signature and return value (reference: openvoice/se_extractor.py:129-152):
``get_se(audio_path, vc_model, target_dir='processed', vad=True) -> (se [1,gin,1], audio_name)``.
The reference first cuts the recording into ~10 s WAV files with a third-party VAD (silero via
whisper_timestamped, se_extractor.py:77-116) or ASR (faster_whisper, :19-74) and then calls
``vc_model.extract_se`` on the pieces.  Those third-party models are out of scope (SURVEY.md
section 2 row 8): here the recording is cut into equal pieces of about ``split_seconds`` -- the same
``num_splits = round(dur / 10)`` rule the reference applies after VAD (se_extractor.py:99-103) --
without removing silence, written under ``target_dir/<name>/wavs`` like the reference, and passed
to ``extract_se``.
"""
import base64
import hashlib
import os
from glob import glob

import numpy as np

from . import audio_io


def hash_numpy_array(audio_path):
    """reference: openvoice/se_extractor.py:118-127 (sha256 of the decoded samples, base64[:16])."""
    try:
        import librosa
        array, _ = librosa.load(audio_path, sr=None, mono=True)
    except ImportError:
        array, _ = audio_io.read_native(audio_path)
    array = np.asarray(array, dtype=np.float32)
    if array.ndim > 1:
        array = array.mean(axis=1)
    digest = hashlib.sha256(np.ascontiguousarray(array).tobytes()).digest()
    return base64.b64encode(digest).decode("utf-8")[:16].replace("/", "_^")


def split_audio_equal(audio_path, audio_name, target_dir, sampling_rate, split_seconds=10.0):
    audio, sr = audio_io.load(audio_path, sr=sampling_rate)
    dur = len(audio) / float(sr)
    wavs_folder = os.path.join(target_dir, audio_name, "wavs")
    os.makedirs(wavs_folder, exist_ok=True)
    num_splits = int(np.round(dur / split_seconds))
    assert num_splits > 0, "input audio is too short"
    bounds = np.linspace(0, len(audio), num_splits + 1).astype(np.int64)
    for i in range(num_splits):
        audio_io.write(os.path.join(wavs_folder, f"{audio_name}_seg{i}.wav"), audio[bounds[i]:bounds[i + 1]], sr)
    return wavs_folder


def get_se(audio_path, vc_model, target_dir="processed", vad=True):
    version = vc_model.version
    print("OpenVoice version:", version)
    audio_name = f"{os.path.basename(audio_path).rsplit('.', 1)[0]}_{version}_{hash_numpy_array(audio_path)}"
    se_path = os.path.join(target_dir, audio_name, "se.pth")
    wavs_folder = split_audio_equal(audio_path, audio_name, target_dir, vc_model.hps.data.sampling_rate)
    audio_segs = sorted(glob(f"{wavs_folder}/*.wav"))
    if len(audio_segs) == 0:
        raise NotImplementedError("No audio segments found!")
    return vc_model.extract_se(audio_segs, se_save_path=se_path), audio_name
