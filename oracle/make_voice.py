"""ORACLE support -- build-container only (needs /root/reference). Writes a Piper voice pair
``<out>.onnx`` + ``<out>.onnx.json`` for a seeded synthetic voice by running the REFERENCE's own
export path: the reference ``SynthesizerTrn`` with ``dec.remove_weight_norm()`` and the
``infer_forward`` wrapper / input names / dynamic axes / opset of
``src/python/piper_train/export_onnx.py:49-101`` through ``torch.onnx.export`` (legacy exporter). The
flow keeps its weight_norm parametrisation, so -- as in real voices -- its conv weights come out as
anonymous constant-folded initialisers. Used to produce the loader fixtures under tests/golden/ and
full-size voices for local experiments:  python oracle/make_voice.py tiny tests/golden/tiny_voice
"""
import contextlib
import json
import os
import sys
import warnings

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_harness as R  # noqa: E402
from piper_amd import weights as W  # noqa: E402


def build_model(preset: str, seed: int):
    """Reference SynthesizerTrn with the seeded canonical weights, dec weight norm removed (export_onnx.py:51-52,
    export_onnx_streaming.py:101-102), the flow still weight-normed as in real voices."""
    cfg = W.preset(preset)
    w = W.synthetic_weights(cfg, seed)
    SynthesizerTrn = R.import_reference()
    with warnings.catch_warnings(), contextlib.redirect_stdout(open(os.devnull, "w")):
        warnings.simplefilter("ignore")
        m = SynthesizerTrn(
            n_vocab=cfg.n_vocab, spec_channels=513, segment_size=32, inter_channels=cfg.inter,
            hidden_channels=cfg.hidden, filter_channels=cfg.filter, n_heads=cfg.n_heads, n_layers=cfg.n_layers,
            kernel_size=cfg.kernel_size, p_dropout=0.1, resblock=str(cfg.resblock),
            resblock_kernel_sizes=cfg.rb_kernel_sizes, resblock_dilation_sizes=cfg.rb_dilations,
            upsample_rates=cfg.up_rates, upsample_initial_channel=cfg.up_initial,
            upsample_kernel_sizes=cfg.up_kernel_sizes, n_speakers=cfg.n_speakers, gin_channels=cfg.gin,
            use_sdp=True).eval()
        m.dec.remove_weight_norm()
    sd = m.state_dict()
    for k, v in w.items():
        t = torch.as_tensor(v)
        if k in sd:
            sd[k] = t
        elif k.endswith(".weight") and k[:-7] + ".weight_v" in sd:     # weight-normed flow layers
            sd[k[:-7] + ".weight_v"] = t
            sd[k[:-7] + ".weight_g"] = t.flatten(1).norm(dim=1).view(-1, 1, 1)
        else:
            raise KeyError(k)
    m.load_state_dict(sd)
    # the legacy exporter's onnxscript post-pass needs the `onnx` package (absent here); it is a no-op
    # for graphs made of stock ops
    import torch.onnx._internal.torchscript_exporter.onnx_proto_utils as opu
    opu._add_onnxscript_fn = lambda model_bytes, custom_opsets: model_bytes
    return cfg, m


def write_config(cfg, preset, seed, path):
    # voice config with the schema of etc/test_voice.onnx.json; a `text` voice over printable ASCII
    pool = " abcdefghijklmnopqrstuvwxyz.,!?'-;:" + "".join(chr(c) for c in range(48, 91))
    chars = list(dict.fromkeys(pool))[: cfg.n_vocab - 3]
    id_map = {"_": [0], "^": [1], "$": [2]}
    id_map.update({ch: [3 + i] for i, ch in enumerate(chars)})
    conf = {"audio": {"sample_rate": cfg.sample_rate}, "espeak": {"voice": "en-us"}, "phoneme_type": "text",
            "inference": {"noise_scale": 0.667, "length_scale": 1, "noise_w": 0.8},
            "phoneme_map": {}, "phoneme_id_map": id_map, "num_symbols": cfg.n_vocab,
            "num_speakers": cfg.n_speakers,
            "speaker_id_map": {f"spk{i}": i for i in range(cfg.n_speakers)} if cfg.n_speakers > 1 else {},
            "synthetic": {"preset": preset, "weight_seed": seed}}
    with open(path, "w", encoding="utf-8") as f:
        json.dump(conf, f, indent=1)


def export_streaming(preset: str, out_dir: str, seed: int = 1234):
    """encoder.onnx + decoder.onnx exactly as the reference's export_onnx_streaming.py:111-190 writes them:
    its own VitsEncoder / VitsDecoder wrapper modules, dummy inputs, names, dynamic axes and opset are used
    (imported from the reference; only the checkpoint loader, which needs pytorch_lightning, is stubbed)."""
    import argparse
    import types
    from pathlib import Path
    cfg, m = build_model(preset, seed)
    name = "piper_train.vits.lightning"
    if name not in sys.modules:
        stub = types.ModuleType(name)
        stub.VitsModel = None
        sys.modules[name] = stub
    from piper_train import export_onnx_streaming as S
    os.makedirs(out_dir, exist_ok=True)
    args = argparse.Namespace(output_dir=Path(out_dir))
    orig_export = torch.onnx.export

    def legacy_export(*a, **kw):
        kw.setdefault("dynamo", False)
        return orig_export(*a, **kw)

    torch.onnx.export = legacy_export
    try:
        torch.manual_seed(1234)                          # export_onnx_streaming.py:73
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            dec_in = S.export_encoder(args, m)
            S.export_decoder(args, m, dec_in)
    finally:
        torch.onnx.export = orig_export
    write_config(cfg, preset, seed, os.path.join(out_dir, "config.json"))
    for f in ("encoder.onnx", "decoder.onnx"):
        print(f"{out_dir}/{f}: {os.path.getsize(os.path.join(out_dir, f))} bytes")


def export(preset: str, out_prefix: str, seed: int = 1234):
    cfg, m = build_model(preset, seed)

    def infer_forward(text, text_lengths, scales, sid=None):            # export_onnx.py:56-69
        return m.infer(text, text_lengths, noise_scale=scales[0], length_scale=scales[1],
                       noise_scale_w=scales[2], sid=sid)[0].unsqueeze(1)

    m.forward = infer_forward
    torch.manual_seed(1234)
    seq = torch.randint(low=0, high=cfg.n_vocab, size=(1, 50), dtype=torch.long)
    lens = torch.LongTensor([50])
    scales = torch.FloatTensor([0.667, 1.0, 0.8])
    sid = torch.LongTensor([0]) if cfg.n_speakers > 1 else None
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        torch.onnx.export(model=m, args=(seq, lens, scales, sid), f=out_prefix + ".onnx", verbose=False,
                          opset_version=15, dynamo=False,
                          input_names=["input", "input_lengths", "scales", "sid"], output_names=["output"],
                          dynamic_axes={"input": {0: "batch_size", 1: "phonemes"},
                                        "input_lengths": {0: "batch_size"},
                                        "output": {0: "batch_size", 1: "time"}})
    write_config(cfg, preset, seed, out_prefix + ".onnx.json")
    print(f"{out_prefix}.onnx: {os.path.getsize(out_prefix + '.onnx')} bytes")


if __name__ == "__main__":
    if sys.argv[1] == "--streaming":     # python oracle/make_voice.py --streaming tiny tests/golden/tiny_stream
        export_streaming(sys.argv[2], sys.argv[3], int(sys.argv[4]) if len(sys.argv) > 4 else 1234)
    else:
        export(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 1234)
