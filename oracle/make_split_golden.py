"""Golden vectors for ``openvoice_amd.utils.split_sentence`` from the UNMODIFIED reference splitter
(/root/reference/openvoice/utils.py:78-194; build container only):  python oracle/make_split_golden.py
Writes tests/golden/split_sentence.json = [{text, min_len, language, pieces}, ...].  ORACLE tooling."""
import importlib.util
import json
import os

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("ref_utils", "/root/reference/openvoice/utils.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

TEXTS = [
    "Did you ever hear a folk tale about a giant turtle? It lived, so they say, in a lake (a deep one) far away.",
    "This is a short one. Ok. Yes! And then a much longer sentence follows, with several clauses, commas; and a semicolon.",
    "He said “hello [EN] world” and <left>. ‘Quoted’ text, «guillemets» too!",
    "One.",
    "A, b, c, d, e, f, g, h, i, j, k, l, m, n, o, p.",
    "No punctuation at all just a long run of words that never stops and goes on and on for a while",
    "今天天气真好，我们一起出去吃饭吧。好的！你想吃什么？我都可以；随便。",
    "Line one\nline two\ttabbed   spaced. Trailing bit",
    "Wait... what?! Really; no. Hm",
    "",
]
out = []
for text in TEXTS:
    for lang in ("EN", "ZH"):
        for min_len in (10, 3, 30):
            out.append(dict(text=text, min_len=min_len, language=lang,
                            pieces=ref.split_sentence(text, min_len=min_len, language_str=lang) if text else None))
out = [o for o in out if o["pieces"] is not None]
with open(os.path.join(REPO, "tests", "golden", "split_sentence.json"), "w", encoding="utf-8") as fh:
    json.dump(out, fh, ensure_ascii=True, indent=0)
print(len(out), "cases")
