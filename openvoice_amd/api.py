"""Public API of the tone-colour converter, re-hosted on the MI355X engine.

Same class names, constructor/ method signatures, attributes (``model``, ``hps``, ``device``,
``version``, ``watermark_model``) and return types as the reference API layer
(reference: openvoice/api.py:14-201), so the upstream notebooks run unchanged
(``from openvoice.api import ToneColorConverter`` resolves here through the ``openvoice`` alias
package at the repo root).  What differs, on purpose:

* the model behind ``self.model`` is ``openvoice_amd.models.SynthesizerTrn`` whose
  ``voice_conversion`` / ``ref_enc`` run hand-written gfx950 kernels -- there is no eager PyTorch
  compute and no CPU fallback (a 'cpu' device is rejected at construction);
* ``ToneColorConverter.__init__`` strips ``enable_watermark`` before calling the base class (the
  reference forwards it and raises TypeError);
* ``convert_batch`` (new) is the batched entry the benchmark configs use; ``convert`` is file I/O
  around ``convert_batch`` with B = 1;
* audio decode / WAV write go through ``openvoice_amd.audio_io`` (librosa/soundfile when present);
* the watermark (third-party ``wavmark``) stays an optional post-processing hook.
"""
import os
import re

import numpy as np
import torch

from . import audio_io, utils
from .mel_processing import spectrogram_torch
from .models import SynthesizerTrn


class OpenVoiceBaseClass(object):
    """reference: openvoice/api.py:14-39."""

    def __init__(self, config_path, device="cuda:0"):
        if "cuda" in device:
            assert torch.cuda.is_available()
        else:
            raise RuntimeError(f"device {device!r}: the MI355X engine has no CPU path; use 'cuda:N'")
        hps = utils.get_hparams_from_file(config_path)
        model = SynthesizerTrn(
            len(getattr(hps, "symbols", [])),
            hps.data.filter_length // 2 + 1,
            n_speakers=hps.data.n_speakers,
            **hps.model,
        ).to(device)
        model.eval()
        self.model = model
        self.hps = hps
        self.device = device

    def load_ckpt(self, ckpt_path):
        checkpoint_dict = torch.load(ckpt_path, map_location=torch.device(self.device))
        a, b = self.model.load_state_dict(checkpoint_dict["model"], strict=False)
        print("Loaded checkpoint '{}'".format(ckpt_path))
        print("missing/unexpected keys:", a, b)


class BaseSpeakerTTS(OpenVoiceBaseClass):
    """V1 base-speaker TTS (reference: openvoice/api.py:42-98).  The model half
    (``SynthesizerTrn.infer``: text encoder, duration predictors, spline flows, flow, generator) runs on
    the MI355X engine.  The TEXT front end of the reference (``openvoice/text``: cleaners built on
    inflect / eng_to_ipa / pypinyin / jieba) is third-party CPU string
    processing and is not re-implemented: plug a ``text_to_sequence(text, symbols, cleaner_names)``
    callable into ``BaseSpeakerTTS.text_to_sequence`` (the reference's own function works unchanged), or
    call ``tts_from_ids`` with symbol ids."""

    language_marks = {"english": "EN", "chinese": "ZH"}
    text_to_sequence = None     # hook: callable(text, symbols, cleaner_names) -> list[int]

    @classmethod
    def get_text(cls, text, hps, is_symbol):
        """reference: openvoice/api.py:48-54."""
        if cls.text_to_sequence is None and is_symbol:
            # already-cleaned (IPA / symbol) text needs no third-party cleaner: plain symbol lookup
            text_norm = utils.cleaned_text_to_sequence(text, hps.symbols)
            if hps.data.add_blank:
                text_norm = intersperse(text_norm, 0)
            return torch.LongTensor(text_norm)
        if cls.text_to_sequence is None:
            raise RuntimeError("no text front end registered: set BaseSpeakerTTS.text_to_sequence to a "
                               "text_to_sequence(text, symbols, cleaner_names) callable (e.g. the reference's "
                               "openvoice.text.text_to_sequence) or use tts_from_ids()")
        text_norm = cls.text_to_sequence(text, hps.symbols, [] if is_symbol else hps.data.text_cleaners)
        if hps.data.add_blank:
            text_norm = intersperse(text_norm, 0)
        return torch.LongTensor(text_norm)

    @staticmethod
    def audio_numpy_concat(segment_data_list, sr, speed=1.0):
        """reference: openvoice/api.py:56-63 -- segments joined with 50 ms / speed of silence after
        each (built with one allocation instead of a Python list of samples)."""
        gap = np.zeros(int((sr * 0.05) / speed), dtype=np.float32)
        parts = []
        for segment in segment_data_list:
            parts += [np.asarray(segment, dtype=np.float32).reshape(-1), gap]
        return np.concatenate(parts) if parts else np.zeros(0, dtype=np.float32)

    @staticmethod
    def split_sentences_into_pieces(text, language_str):
        """reference: openvoice/api.py:65-71 (prints the pieces like the reference does)."""
        texts = utils.split_sentence(text, language_str=language_str)
        print(" > Text splitted to sentences.")
        print("\n".join(texts))
        print(" > ===========================")
        return texts

    @torch.no_grad()
    def tts_from_ids(self, id_sequences, speaker_id, speed=1.0, batched=False, noise_scale=0.667,
                     noise_scale_w=0.6):
        """Synthesize already-tokenised sentences (symbol ids, blanks interspersed by the caller if the
        config asks for it).  ``batched=False`` runs one ``infer`` per sentence exactly as the reference loop
        (api.py:78-94); ``batched=True`` pads them into one batch (one pass over the GPU; because the
        generator is unmasked, the last ~13 frames of the shorter items then differ slightly from a
        per-sentence run).  Returns a list of float32 numpy waveforms."""
        device = self.device
        seqs = [torch.as_tensor(s, dtype=torch.long).reshape(-1) for s in id_sequences]
        hop = self.hps.data.hop_length
        if not batched:
            out = []
            for s in seqs:
                o = self.model.infer(s[None].to(device), torch.LongTensor([s.numel()]).to(device),
                                     sid=torch.LongTensor([speaker_id]).to(device), noise_scale=noise_scale,
                                     noise_scale_w=noise_scale_w, length_scale=1.0 / speed)[0]
                out.append(o[0, 0].data.cpu().float().numpy())
            return out
        lengths = torch.tensor([s.numel() for s in seqs], dtype=torch.long)
        x = torch.zeros(len(seqs), int(lengths.max()), dtype=torch.long)
        for i, s in enumerate(seqs):
            x[i, :s.numel()] = s
        sid = torch.full((len(seqs),), int(speaker_id), dtype=torch.long)
        # padded batch: the generator computes length + 16 frames per sentence, not the longest one's (the samples
        # returned below are bit-identical either way)
        o, _, y_mask, _ = self.model.infer(x.to(device), lengths.to(device), sid=sid.to(device), noise_scale=noise_scale,
                                           noise_scale_w=noise_scale_w, length_scale=1.0 / speed, skip_padding=True)
        frames = y_mask[:, 0].sum(1).long().cpu().tolist()
        o = o[:, 0].data.cpu().float().numpy()
        return [o[i, :frames[i] * hop] for i in range(len(seqs))]

    def tts(self, text, output_path, speaker, language="English", speed=1.0, batched=False):
        """reference: openvoice/api.py:73-98."""
        mark = self.language_marks.get(language.lower(), None)
        assert mark is not None, f"language {language} is not supported"
        texts = self.split_sentences_into_pieces(text, mark)
        ids = []
        for t in texts:
            t = re.sub(r"([a-z])([A-Z])", r"\1 \2", t)
            t = f"[{mark}]{t}[{mark}]"
            ids.append(self.get_text(t, self.hps, False))
        audio_list = self.tts_from_ids(ids, self.hps.speakers[speaker], speed=speed, batched=batched)
        audio = self.audio_numpy_concat(audio_list, sr=self.hps.data.sampling_rate, speed=speed)
        if output_path is None:
            return audio
        audio_io.write(output_path, audio, self.hps.data.sampling_rate)


def intersperse(lst, item):
    """reference: openvoice/commons.py:22-25."""
    result = [item] * (len(lst) * 2 + 1)
    result[1::2] = lst
    return result


def string_to_bits(string, pad_len=8):
    """ASCII -> [pad_len, 8] bit matrix, MSB first, short strings padded with 0b00100000 (space);
    reference: openvoice/utils.py:46-62."""
    codes = np.frombuffer(string.encode("latin-1", "replace")[:pad_len], dtype=np.uint8)
    full = np.full(pad_len, 0x20, dtype=np.uint8)
    full[:len(codes)] = codes
    return np.unpackbits(full[:, None], axis=1).astype(np.int64)


def bits_to_string(bits_array):
    """reference: openvoice/utils.py:65-75."""
    vals = np.packbits(np.asarray(bits_array).astype(np.uint8), axis=1).reshape(-1)
    return "".join(chr(int(v)) for v in vals)


class ToneColorConverter(OpenVoiceBaseClass):
    """reference: openvoice/api.py:101-201."""

    def __init__(self, *args, **kwargs):
        enable_watermark = kwargs.pop("enable_watermark", True)
        super().__init__(*args, **kwargs)
        self.watermark_model = None
        if enable_watermark:
            try:
                import wavmark   # third-party, not part of the reference tree (requirements.txt:4)
                self.watermark_model = wavmark.load_model().to(self.device)
            except ImportError:
                print("wavmark is not installed: watermarking disabled")
        self.version = getattr(self.hps, "_version_", "v1")
        self.use_graphs = False

    def enable_graphs(self, enable=True):
        """Replay conversions of an already-seen (batch, frames, tau) shape from a captured HIP graph: one
        ``hipGraphLaunch`` instead of ~330 launches through Python.  Pays at small batches (a single file is
        launch-bound in eager mode); each distinct shape costs one capture and keeps its workspace resident
        (the 4 most recent shapes are kept).  Off by default."""
        self.use_graphs = bool(enable)
        return self

    def enable_split_bf16x3(self, enable=True, products=6):
        """Run the generator's MRF stages with C >= 64 on the split-precision kernels (``ConverterEngine.use_split_bf16x3``:
        three bf16 planes per fp32 operand, six plane products, fp32 accumulation -- fp32-level results on the bf16 matrix
        pipe, 1.3x the fp32 path's speed on a batch).  Off by default; call after ``load_ckpt``
        (the engine is rebuilt when the weights change)."""
        self.model.engine().use_split_bf16x3(enable, products=products)
        return self

    # ---- spectrogram helpers ---------------------------------------------------------------------
    def _spec(self, y):
        d = self.hps.data
        return spectrogram_torch(y, d.filter_length, d.sampling_rate, d.hop_length, d.win_length, center=False)

    def extract_se(self, ref_wav_list, se_save_path=None):
        """Mean reference-encoder embedding over the given audio files -> ``[1, gin, 1]``
        (reference: openvoice/api.py:114-139, which runs one spectrogram + one ``ref_enc`` call per file).  Here the
        files of equal length -- the ~10 s pieces ``se_extractor.get_se`` cuts a recording into -- are stacked into ONE
        ``[N, samples]`` spectrogram launch pair and ONE ``ref_enc`` launch sequence (the kernels take N); files of
        other lengths keep the per-file path.  The mean runs over files in the caller's order either way."""
        if isinstance(ref_wav_list, str):
            ref_wav_list = [ref_wav_list]
        # (decoded on the host, resampled -- where the file's rate differs -- by the device kernel: audio_io.load_to_device)
        audios = [audio_io.load_to_device(f, self.hps.data.sampling_rate, self.device) for f in ref_wav_list]
        gs = self.extract_se_from_audio(audios)
        if se_save_path is not None:
            os.makedirs(os.path.dirname(se_save_path), exist_ok=True)
            torch.save(gs.cpu(), se_save_path)
        return gs

    @torch.no_grad()
    def extract_se_from_audio(self, audios):
        """``extract_se`` on decoded float32 waveforms (list of 1-D arrays / tensors at the model's sampling rate)."""
        groups = {}
        for i, a in enumerate(audios):
            groups.setdefault(len(a), []).append(i)
        embs = [None] * len(audios)
        self.last_extract_se_batches = []          # (files in the ref_enc call) per call, for tests / logs
        for length, idx in groups.items():
            y = torch.stack([torch.as_tensor(audios[i], dtype=torch.float32) for i in idx]).to(self.device)
            spec = self._spec(y)                                             # [n, 513, T], one launch pair
            g = self.model.ref_enc(spec.transpose(1, 2))                     # [n, gin], one launch sequence
            self.last_extract_se_batches.append(len(idx))
            for row, i in enumerate(idx):
                embs[i] = g[row]
        return torch.stack(embs).mean(0).reshape(1, -1, 1).detach()

    @torch.no_grad()
    def convert_batch(self, waveforms, src_se, tgt_se, tau=0.3, noise=None):
        """Batched conversion, everything on the device.

        ``waveforms``: float32 tensor ``[B, N]`` (equal lengths), or a list of 1-D tensors/arrays of
        different lengths (each gets its own spectrogram, exactly as a per-file ``convert`` would,
        then the batch is zero-padded in frames and masked by ``spec_lengths``).  Returns
        ``(o_hat [B, 1, hop*T_max] on the device, lengths_in_samples [B])``.  Note the reference
        decoder is unmasked, so for ragged batches samples within ~13 frames of an utterance's end
        differ from a per-utterance run (hard part 6); trim with the returned
        lengths."""
        hop = self.hps.data.hop_length
        ragged = isinstance(waveforms, (list, tuple))
        if ragged:
            specs = [self._spec(torch.as_tensor(w, dtype=torch.float32).to(self.device).reshape(1, -1))[0]
                     for w in waveforms]
            frames = [s.shape[1] for s in specs]
            spec = torch.zeros(len(specs), specs[0].shape[0], max(frames), dtype=torch.float32, device=self.device)
            for i, s in enumerate(specs):
                spec[i, :, :s.shape[1]] = s
            spec_lengths = torch.tensor(frames, dtype=torch.int64, device=self.device)
        else:
            y = torch.as_tensor(waveforms, dtype=torch.float32).to(self.device)
            spec = self._spec(y)
            spec_lengths = torch.full((spec.shape[0],), spec.shape[2], dtype=torch.int64, device=self.device)
        if self.use_graphs:
            # the graph's outputs are static buffers: hand the caller its own copy
            o_hat = self.model.voice_conversion(spec, spec_lengths, sid_src=src_se, sid_tgt=tgt_se, tau=tau,
                                                noise=noise, graph=True, skip_padding=ragged)[0].clone()
        else:
            # ragged batch: the generator skips what lies beyond length + 16 frames of each utterance (the samples
            # within the returned lengths are bit-identical to the full computation; the padded tail is zero)
            o_hat = self.model.voice_conversion(spec, spec_lengths, sid_src=src_se, sid_tgt=tgt_se, tau=tau,
                                                noise=noise, skip_padding=ragged)[0]
        return o_hat, spec_lengths * hop

    @torch.no_grad()
    def convert_batch_sharded(self, waveforms, src_se, tgt_se, tau=0.3, noise=None, gather=True):
        """``convert_batch`` across the GPUs of one node: call it from every rank of an
        initialised ``torch.distributed`` process group (backend "nccl" = RCCL; one process per GPU, this converter on
        the rank's own device) with the same ``waveforms`` -- a ``[N, samples]`` tensor, or a list of N waveforms of
        different lengths.  Rank 0's speaker embeddings are broadcast (2 KiB, the only collective on the path; other
        ranks may pass None), rank r converts utterances ``parallel.shard_range(N, r, world)``, and with ``gather``
        every rank receives the whole ``[N, 1, hop*T_max]`` batch (one all-gather), else ``(local_o_hat, (start,
        end))``.  ``noise`` [N, 192, T] is per utterance, so the result does not depend on the number of ranks.  For a
        list the return value is ``(o_hat, lengths_in_samples [N])`` as from ``convert_batch`` (every rank knows every
        length, so the padded width is agreed on without a collective).  Without a process group this is
        ``convert_batch``."""
        from . import parallel
        gin = self.model.model_cfg["gin_channels"]
        d = self.hps.data
        hop = d.hop_length
        frames_of = lambda n: (int(n) + (d.filter_length - hop) // 2 * 2 - d.filter_length) // hop + 1
        ragged = isinstance(waveforms, (list, tuple))
        if ragged:
            frames = [frames_of(len(w)) for w in waveforms]
            items = list(waveforms)
        else:
            items = torch.as_tensor(waveforms, dtype=torch.float32)
            frames = [frames_of(items.shape[1])] * len(items)
        width = max(frames) * hop                        # samples of the padded batch, the same on every rank

        def convert(shard, s, t, nz):
            if len(shard) == 0:         # more ranks than utterances: an empty shard of the agreed width
                return torch.zeros(0, 1, width, dtype=torch.float32, device=self.device)
            # (nz arrives cut to the shard's own longest length: parallel.convert_sharded(frames=...))
            o = self.convert_batch(shard, s, t, tau=tau, noise=nz)[0]
            if o.shape[2] < width:      # a shard whose longest utterance is shorter than the batch's
                o = torch.nn.functional.pad(o, (0, width - o.shape[2]))
            return o
        out = parallel.convert_sharded(convert, items, src_se, tgt_se, gin, self.device, noise=noise, gather=gather,
                                       frames=frames)
        if ragged and gather:
            return out, torch.tensor(frames, dtype=torch.int64, device=self.device) * hop
        return out

    def convert(self, audio_src_path, src_se, tgt_se, output_path=None, tau=0.3, message="default"):
        """reference: openvoice/api.py:141-160."""
        hps = self.hps
        y = audio_io.load_to_device(audio_src_path, hps.data.sampling_rate, self.device).unsqueeze(0)
        o_hat, _ = self.convert_batch(y, src_se, tgt_se, tau=tau)
        audio = o_hat[0, 0].data.cpu().float().numpy()
        audio = self.add_watermark(audio, message)
        if output_path is None:
            return audio
        audio_io.write(output_path, audio, hps.data.sampling_rate)

    # ---- optional watermark hook (third-party model; behaviour of the reference's openvoice/api.py:162-201) ----------
    # The message travels as 32-bit groups, group n in the 16 000-sample window that starts at sample 32 000 n (every other
    # window of the waveform stays untouched); ``watermark_model`` is any object with ``encode(signal [1, 16000], bits
    # [1, 32]) -> signal`` and ``decode(signal [1, 16000]) -> scores [1, 32]`` (wavmark's interface).
    WATERMARK_WINDOW = 16000
    WATERMARK_STRIDE = 2 * WATERMARK_WINDOW
    WATERMARK_GROUP_BITS = 32

    def _watermark_windows(self, audio, count):
        """(n, start, stop) of the first ``count`` carrier windows that lie wholly inside ``audio``; stops at the first
        that does not."""
        for n in range(count):
            start = n * self.WATERMARK_STRIDE
            if start + self.WATERMARK_WINDOW > len(audio):
                return
            yield n, start, start + self.WATERMARK_WINDOW

    def add_watermark(self, audio, message):
        """Embed ``message`` (padded / cut to 8 latin-1 characters) into ``audio`` (1-D float array, modified in place and
        returned).  A waveform too short for all groups carries the leading ones and a notice is printed, as upstream."""
        if self.watermark_model is None:
            return audio
        payload = string_to_bits(message).reshape(-1)
        groups = len(payload) // self.WATERMARK_GROUP_BITS
        done = 0
        with torch.no_grad():
            for n, start, stop in self._watermark_windows(audio, groups):
                carrier = torch.as_tensor(audio[start:stop], dtype=torch.float32, device=self.device).unsqueeze(0)
                bits = torch.as_tensor(payload[n * self.WATERMARK_GROUP_BITS:(n + 1) * self.WATERMARK_GROUP_BITS],
                                       dtype=torch.float32, device=self.device).unsqueeze(0)
                audio[start:stop] = self.watermark_model.encode(carrier, bits).detach().cpu().reshape(-1).numpy()
                done += 1
        if done < groups:
            print("Audio too short, fail to add watermark")
        return audio

    def detect_watermark(self, audio, n_repeat):
        """Read ``n_repeat`` 32-bit groups back (threshold 0.5 on the model's scores) and decode them to characters;
        the string "Fail" when the waveform does not hold that many carrier windows."""
        groups = []
        with torch.no_grad():
            for _, start, stop in self._watermark_windows(audio, n_repeat):
                carrier = torch.as_tensor(audio[start:stop], dtype=torch.float32, device=self.device).unsqueeze(0)
                scores = self.watermark_model.decode(carrier)
                groups.append((scores >= 0.5).to(torch.uint8).detach().cpu().reshape(-1).numpy())
        if len(groups) < n_repeat:
            print("Audio too short, fail to detect watermark")
            return "Fail"
        return bits_to_string(np.stack(groups).reshape(-1, 8))
