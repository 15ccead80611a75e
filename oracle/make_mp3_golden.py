#!/usr/bin/env python3
"""Golden vectors for openvoice_amd/mp3.py (the from-scratch MPEG-1 Layer III decoder behind ``audio_io.load``;
reference call sites: openvoice/api.py:123,144 ``librosa.load(path)`` on resources/*.mp3 = BASELINE.json configs[0]).

Test infrastructure, build container only.  The image holds no audio library, but the `kaleido` package bundles a headless
Chromium whose WebAudio ``decodeAudioData`` runs FFmpeg's MP3 decoder.  This script hands kaleido a stand-in "plotly.js"
whose ``toImage`` decodes the MP3 carried in the figure (at the file's own sampling rate, so nothing is resampled) and
returns the PCM; from it small fixtures are committed under tests/golden/:

    mp3_<name>.npz: rate, channels, samples (what Chromium returns after its gapless trimming), three 8192-sample
                    excerpts per channel (start / middle / end) with their offsets, the RMS of every 1152-sample block of
                    the whole file, and sha256 of the MP3 the vectors belong to.

``invalid_keypress.mp3`` ships inside the image itself (kaleido's MathJax), so its test runs everywhere; the reference's
resources/*.mp3 exist only where /root/reference does, and their tests skip elsewhere.

    python oracle/make_mp3_golden.py"""
import base64
import hashlib
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYPRESS = "/usr/local/lib/python3.10/dist-packages/kaleido/executable/etc/mathjax/extensions/a11y/invalid_keypress.mp3"
FILES = [KEYPRESS] + [os.path.join("/root/reference/resources", n) for n in
                      ("example_reference.mp3", "demo_speaker0.mp3", "demo_speaker1.mp3", "demo_speaker2.mp3")]

FAKE_PLOTLY = r"""
window.Plotly = {
  version: '2.0.0',
  purge: function () {},
  toImage: function (fig, opts) {
    return new Promise(function (resolve) {
      try {
        var spec = fig.data[0];
        var bin = atob(spec.mp3_base64);
        var buf = new Uint8Array(bin.length);
        for (var i = 0; i < bin.length; i++) buf[i] = bin.charCodeAt(i);
        var Ctx = window.OfflineAudioContext || window.webkitOfflineAudioContext;
        var ctx = new Ctx(spec.channels, 1, spec.sample_rate);
        ctx.decodeAudioData(buf.buffer, function (audio) {
          var out = [];
          var head = audio.numberOfChannels + ',' + audio.length + ',' + audio.sampleRate;
          for (var c = 0; c < audio.numberOfChannels; c++) {
            var f32 = audio.getChannelData(c);
            var u8 = new Uint8Array(f32.buffer, f32.byteOffset, f32.byteLength);
            var s = '';
            for (var k = 0; k < u8.length; k += 8192) s += String.fromCharCode.apply(null, u8.subarray(k, k + 8192));
            out.push(btoa(s));
          }
          resolve(head + '|' + out.join('|'));
        }, function (err) { resolve('ERROR decode: ' + err); });
      } catch (e) { resolve('ERROR ' + e); }
    });
  }
};
"""


def chromium_decode(mp3_bytes, channels, rate, scope):
    fig = {"data": [{"mp3_base64": base64.b64encode(mp3_bytes).decode(), "channels": channels, "sample_rate": rate}],
           "layout": {}}
    resp = scope._perform_transform(fig, format="svg", width=100, height=100, scale=1)
    text = resp.get("result") or ""
    if resp.get("code") != 0 or text.startswith("ERROR"):
        raise RuntimeError(f"chromium decode failed: {resp.get('code')} {resp.get('message')} {text[:200]}")
    head, *chans = text.split("|")
    nch, n, sr = (int(float(v)) for v in head.split(","))
    pcm = np.stack([np.frombuffer(base64.b64decode(c), dtype="<f4") for c in chans])
    assert pcm.shape == (nch, n) and sr == rate, (pcm.shape, nch, n, sr)
    return pcm


def main():
    sys.path.insert(0, REPO)
    from kaleido.scopes.plotly import PlotlyScope
    from openvoice_amd import mp3
    js = os.path.join("/tmp", "ov_fake_plotly.js")
    with open(js, "w") as fh:
        fh.write(FAKE_PLOTLY)
    scope = PlotlyScope(plotlyjs=js)
    out_dir = os.path.join(REPO, "tests", "golden")
    for path in FILES:
        if not os.path.exists(path):
            print("skip (absent):", path)
            continue
        data = open(path, "rb").read()
        info = mp3.probe(data)
        pcm = chromium_decode(data, info["channels"], info["sample_rate"], scope)
        n = pcm.shape[1]
        offs = [0, max(0, n // 2 - 4096), max(0, n - 8192)]
        nb = n // 1152
        rms = np.sqrt((pcm[:, :nb * 1152].reshape(pcm.shape[0], nb, 1152).astype(np.float64) ** 2).mean(-1)).astype(np.float32)
        name = os.path.splitext(os.path.basename(path))[0]
        np.savez_compressed(os.path.join(out_dir, f"mp3_{name}.npz"), rate=info["sample_rate"], channels=pcm.shape[0],
                            samples=n, offsets=np.array(offs), excerpts=np.stack([pcm[:, o:o + 8192] for o in offs]),
                            block_rms=rms, sha256=hashlib.sha256(data).hexdigest(), source=path,
                            full=pcm if n <= 40000 else np.zeros((0,), np.float32))
        print(name, info, "->", pcm.shape, "peak", float(np.abs(pcm).max()))


if __name__ == "__main__":
    main()
