"""Config plumbing for the converter boundary.

Mirrors the reference's JSON -> nested attribute-bag loader
(reference: openvoice/utils.py:6-43 ``get_hparams_from_file`` / ``HParams``) so
``ToneColorConverter(config_path, device)`` accepts the released ``config.json``
files unchanged.  Only the config loader is in scope;
the watermark bit codec and the sentence splitters are not on the converter path.
"""
import json


class HParams:
    """Nested attribute bag: ``hps.data.sampling_rate`` and ``hps['data']`` both work,
    and ``**hps.model`` expands (needs ``keys`` + ``__getitem__``), as the reference's
    constructor call does (reference: openvoice/api.py:23-28)."""

    def __init__(self, **kwargs):
        for key, value in kwargs.items():
            self[key] = HParams(**value) if isinstance(value, dict) else value

    def keys(self):
        return self.__dict__.keys()

    def items(self):
        return self.__dict__.items()

    def values(self):
        return self.__dict__.values()

    def __len__(self):
        return len(self.__dict__)

    def __getitem__(self, key):
        return getattr(self, key)

    def __setitem__(self, key, value):
        setattr(self, key, value)

    def __contains__(self, key):
        return key in self.__dict__

    def __repr__(self):
        return repr(self.__dict__)


def get_hparams_from_file(config_path):
    with open(config_path, "r", encoding="utf-8") as handle:
        return HParams(**json.load(handle))


def split_sentence(text, min_len=10, language_str="[EN]"):
    """Sentence pieces for ``BaseSpeakerTTS.tts`` -- the reference's policy (openvoice/utils.py:78-194), checked piece
    for piece against the reference splitter in tests/test_api_cpu.py (golden: tests/golden/split_sentence.json):

    1. full-width punctuation -> ASCII; Latin text ('EN') also loses quotes and ``<>()[]"`` brackets, so user text
       can never inject the ``[EN]`` language marks ``BaseSpeakerTTS.tts`` wraps around each piece;
    2. cut after every ``, . ! ? ;``;
    3. glue consecutive cuts together until a piece exceeds ``min_len`` words (Latin) / characters (anything else);
    4. a piece of <= 2 words / characters is merged into its successor, a trailing one into its predecessor.

    Only ``language_str == 'EN'`` takes the Latin rules, like the reference (utils.py:79)."""
    import re
    latin = language_str == "EN"
    size = (lambda piece: len(piece.split(" "))) if latin else len
    for pattern, repl in (("[\u3002\uff01\uff1f\uff1b]", "."), ("[\uff0c]", ",")):
        text = re.sub(pattern, repl, text)
    if latin:
        text = re.sub("[\u201c\u201d]", '"', text)
        text = re.sub("[\u2018\u2019]", "'", text)
        text = re.sub(r'[<>()\[\]"\u00ab\u00bb]+', "", text)
    text = re.sub("[\n\t ]+", " ", text)
    cuts = [c.strip() for c in re.sub("([,.!?;])", "\\1 \x00", text).split("\x00")]
    if cuts and not cuts[-1]:
        cuts.pop()
    pieces, run, count = [], [], 0
    for n, cut in enumerate(cuts):
        run.append(cut)
        count += size(cut)
        if count > min_len or n == len(cuts) - 1:
            pieces.append(" ".join(run))
            run, count = [], 0
    merged = []
    for piece in pieces:
        if merged and size(merged[-1]) <= 2:
            merged[-1] += " " + piece
        else:
            merged.append(piece)
    if len(merged) > 1 and size(merged[-1]) <= 2:
        last = merged.pop()
        merged[-1] += " " + last
    return merged


# Hyper-parameters of the released converter checkpoints (tag [K]).
# Used by bench.py / tests when no config.json is supplied; never hard-wired in the engine.
CONVERTER_MODEL_CONFIG = dict(
    inter_channels=192,
    hidden_channels=192,
    filter_channels=768,
    n_heads=2,
    n_layers=6,
    kernel_size=3,
    p_dropout=0.1,
    resblock="1",
    resblock_kernel_sizes=[3, 7, 11],
    resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5], [1, 3, 5]],
    upsample_rates=[8, 8, 2, 2],
    upsample_initial_channel=512,
    upsample_kernel_sizes=[16, 16, 4, 4],
    gin_channels=256,
)

CONVERTER_DATA_CONFIG = dict(
    sampling_rate=22050,
    filter_length=1024,
    hop_length=256,
    win_length=1024,
    n_speakers=0,
)


def default_converter_hparams(version="v2"):
    """An ``HParams`` equal to what ``get_hparams_from_file`` yields for the released
    converter ``config.json`` (V2 adds ``_version_`` and ``zero_g``; reference:
    openvoice/api.py:110, openvoice/models.py:423,465,495,498)."""
    model = dict(CONVERTER_MODEL_CONFIG)
    cfg = dict(data=dict(CONVERTER_DATA_CONFIG), model=model)
    if version == "v2":
        cfg["_version_"] = "v2"
        model["zero_g"] = True
    return HParams(**cfg)


def cleaned_text_to_sequence(cleaned_text, symbols):
    """Symbol ids of an already-cleaned string; characters outside ``symbols`` are dropped
    (reference: openvoice/text/__init__.py:34-43)."""
    symbol_to_id = {s: i for i, s in enumerate(symbols)}
    return [symbol_to_id[ch] for ch in cleaned_text if ch in symbol_to_id]
