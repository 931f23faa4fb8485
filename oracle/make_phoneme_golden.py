"""ORACLE support -- build-container only. Extracts the reference's own known-answer data for the
phonemes -> ids step that feeds the hot path: rows of
/root/reference/etc/test_sentences/test_en-us.jsonl (phonemes + phoneme_ids produced by
piper-phonemize) and the phoneme_id_map of /root/reference/etc/test_voice.onnx.json, into
tests/golden/phoneme_ids_en-us.json."""
import json
import os

REF = "/root/reference/etc"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

rows = [json.loads(l) for l in open(os.path.join(REF, "test_sentences", "test_en-us.jsonl"), encoding="utf-8")]
conf = json.load(open(os.path.join(REF, "test_voice.onnx.json"), encoding="utf-8"))
out = {"source": "reference etc/test_sentences/test_en-us.jsonl + etc/test_voice.onnx.json",
       "phoneme_id_map": conf["phoneme_id_map"],
       "rows": [{"phonemes": r["phonemes"], "phoneme_ids": r["phoneme_ids"]} for r in rows]}
with open(os.path.join(ROOT, "tests", "golden", "phoneme_ids_en-us.json"), "w", encoding="utf-8") as f:
    json.dump(out, f, ensure_ascii=False)
print(len(rows), "rows", [len(r["phoneme_ids"]) for r in rows])
