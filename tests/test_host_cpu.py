"""CPU: product host logic (no GPU compute) against the golden vectors captured from the reference,
and the C-ABI library's exported symbols against include/ufv.h."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import load_golden, t, ROOT
from ufvideo_amd import mm_utils as MU
from ufvideo_amd import constants as C
from ufvideo_amd.model.videorefer_arch import build_splice_plan, splice_labels_and_mask, splice_index_arrays


class CharTok:
    bos_token_id = None

    def __call__(self, text, add_special_tokens=False):
        r = type("R", (), {})()
        r.input_ids = [ord(c) for c in text]
        return r


def test_constants_match_reference_values():
    assert (C.IGNORE_INDEX, C.IMAGE_TOKEN_INDEX, C.VIDEO_TOKEN_INDEX, C.AUDIO_TOKEN_INDEX) == (-100, -200, -201, -202)
    assert C.NUM_FRAMES == C.MAX_FRAMES == 32 and C.TEMPORAL_TOKEN_FORMAT.format(7) == "<TEMP-007>"
    assert C.MODAL_INDEX_MAP == {"<image>": -200, "<video>": -201, "<audio>": -202}


def test_frame_sample_kats():
    a, _ = load_golden("int_helpers")
    n = 0
    for k in a:
        if k.startswith("uniform_"):
            _, d, nf = k.split("_")
            assert (MU.frame_sample(int(d), "uniform", int(nf)) == a[k]).all(), k; n += 1
        elif k.startswith("fps_"):
            _, d, fps = k.split("_")
            assert (MU.frame_sample(int(d), "fps", fps=float(fps)) == a[k]).all(), k; n += 1
    assert n >= 10
    with pytest.raises(ImportError):
        MU.frame_sample(10, mode="bogus")
    with pytest.raises(AssertionError):
        MU.frame_sample(10, mode="uniform")


def test_tokenizer_multimodal_token_and_names():
    a, _ = load_golden("int_helpers")
    tok = CharTok()
    for i in range(5):
        mt, prompt = bytes(a[f"tokprompt_{i}"]).decode().split("|", 1)
        assert MU.tokenizer_multimodal_token(prompt, tok, mt, return_tensors="pt").tolist() == a[f"tok_{i}"].tolist()
    assert MU.tokenizer_multimodal_token("plain", tok, "") == a["tok_plain"].tolist()
    with pytest.raises(ValueError):
        MU.tokenizer_multimodal_token("x", tok, "<video>", return_tensors="np")
    assert MU.get_model_name_from_path("/a/b/UFVideo-7B/") == bytes(a["model_name_a"]).decode()
    assert MU.get_model_name_from_path("/a/run1/checkpoint-300") == bytes(a["model_name_b"]).decode()


def test_expand2square_and_sam_preprocess():
    from PIL import Image
    a, _ = load_golden("int_helpers")
    for i in range(3):
        out = np.array(MU.expand2square(Image.fromarray(a[f"sq_in_{i}"]), (127, 127, 127)))
        assert np.array_equal(out, a[f"sq_out_{i}"])
    assert torch.allclose(MU.sam_preprocess(t(a["sam_pre_in"])), t(a["sam_pre_out"]))


def test_image_processor_matches_hf_siglip_processor():
    a, _ = load_golden("processor")
    p = MU.UfvImageProcessor(size=56)
    for i in range(3):
        o = p.preprocess([a[f"in_{i}"]])["pixel_values"][0]
        assert (o - t(a[f"out_{i}"])).abs().max() < 1e-6
    assert p.image_mean == [0.5, 0.5, 0.5] and p.size == {"height": 56, "width": 56}


def test_process_video_tail_pads_and_truncates():
    p = MU.UfvImageProcessor(size=28)
    rng = np.random.default_rng(0)
    frames = [rng.integers(0, 256, (20, 30, 3), dtype=np.uint8) for _ in range(3)]
    video, fd, h, w, fl = MU.process_video(frames, p, aspect_ratio="square", num_frames=5, frame_idx=[0, 2])
    assert video.shape == (5, 3, 28, 28) and fd.shape == (2, 3, 28, 28) and (h, w) == (20, 30) and len(fl) == 2
    assert torch.all(video[3:] == -1.0)                       # black padding frames -> (0/255-0.5)/0.5
    video2, fd2, *_ = MU.process_video(np.stack(frames), p, aspect_ratio="pad", num_frames=2)
    assert video2.shape == (2, 3, 28, 28) and fd2 is None
    with pytest.raises(ValueError):
        MU.process_video(3.14, p)


def test_keywords_stopping_criteria():
    class Tok(CharTok):
        def batch_decode(self, ids, skip_special_tokens=True):
            return ["".join(chr(int(i)) for i in row) for row in ids]
    crit = MU.KeywordsStoppingCriteria(["</s>"], Tok(), torch.zeros(1, 3, dtype=torch.long))
    ids = lambda s: torch.tensor([[ord(c) for c in s]])
    assert crit(ids("abc </s>")) and not crit(ids("abc </s"))


CASES = {"vid_region": True, "vid_only": False, "img_only": False, "batch_pad": False, "vid_noregion_frame": True,
         "vid_trailing": False}


def test_splice_plan_bit_exact_against_reference():
    """mark / attention mask / labels / embedding row placement for every golden splice case."""
    a, w = load_golden("model_tiny")
    R = int(a["region_id"])
    table = w["model.embed_tokens.weight"]
    mm = t(a["mm_features"])[0]                          # features of `video`
    for name, have_frame in CASES.items():
        ids = t(a[f"sp_{name}_ids"]); am = t(a[f"sp_{name}_am_in"])
        n_mm = int((ids < 0).sum())
        nums = {"vid_region": [1, 1], "vid_noregion_frame": [1]}.get(name, [])
        mm_lens = [mm.shape[0]] * n_mm
        plan = build_splice_plan(ids.tolist(), mm_lens, nums, R, have_frame)
        for lab in (False, True):
            labels = None
            if lab:
                labels = ids.clone(); labels[labels < 0] = -100
            tag = f"{name}_{'lab' if lab else 'nolab'}"
            nl, nm = splice_labels_and_mask(plan, ids, am, labels, mm_lens)
            assert np.array_equal(np.array(plan.mark), a[f"sp_{tag}_mark"]), tag
            assert np.array_equal(nm.numpy(), a[f"sp_{tag}_am"]), tag
            if lab:
                assert np.array_equal(nl.numpy(), a[f"sp_{tag}_labels"]), tag
        # text rows land where the reference put embed_tokens rows
        (ts, td), (ms, md), (rs, rd) = splice_index_arrays(plan, ids.tolist(), mm_lens, [k * mm.shape[0] for k in range(n_mm)])
        emb = t(a[f"sp_{name}_nolab_emb"]); S = emb.shape[1]
        flat = emb.reshape(-1, emb.shape[-1])
        assert torch.equal(flat[td], table[ts]), name
        assert len(ts) + len(ms) + len(rs) == sum(plan.lengths)
        if name in ("vid_only", "vid_trailing", "vid_region"):
            assert torch.allclose(flat[md], mm[ms], atol=1e-6), name


def test_c_abi_exports_every_declared_symbol():
    from ufvideo_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "ufv.h")).read()
    declared = set(re.findall(r"\b(ufv_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 18
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/ufv.h but not exported"
    assert set(_lib.SIGNATURES) | {"ufv_last_error", "ufv_abi_version"} | set(_lib.SIZE_FUNCS) == declared
    # ... and NOTHING else: -fvisibility=hidden + UFV_API + csrc/ufv_exports.map (no mangled internals, no compiler markers in the dynamic ABI)
    import shutil
    import subprocess
    nm = shutil.which("nm") or "/opt/rocm/lib/llvm/bin/llvm-nm"
    out = subprocess.run([nm, "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if l.strip()}
    assert exported == declared, (sorted(exported - declared)[:10], sorted(declared - exported)[:10])
    assert all(re.search(r"UFV_API [a-z_0-9\* ]*\b%s\(" % n, hdr) for n in declared)       # every declaration carries the export macro
    assert _lib.load().ufv_abi_version() == _lib.ABI_VERSION == 3


def test_library_path_cannot_be_swapped_by_the_environment_outside_lab_runs():
    """UFV_LIBRARY alone is ignored (and said so); only UFV_LAB=1 UFV_LIBRARY=... loads another build (tools/lab/ab_bench.sh)"""
    import subprocess
    import sys
    code = "from ufvideo_amd import _lib; print(_lib.LIB_PATH)"
    env = {k: v for k, v in os.environ.items() if k not in ("UFV_LAB", "UFV_LIBRARY")}
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=dict(env, UFV_LIBRARY="/tmp/other.so"), capture_output=True, text=True)
    assert r.stdout.strip().endswith("ufvideo_amd/libufv_hip.so") and "ignored" in r.stderr
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=dict(env, UFV_LIBRARY="/tmp/other.so", UFV_LAB="1"), capture_output=True, text=True)
    assert r.stdout.strip() == "/tmp/other.so" and "LAB RUN" in r.stderr


def test_ops_fail_loudly_without_gpu_tensors():
    from ufvideo_amd import ops, _lib
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.UfvError):
        ops.gemm(torch.zeros(4, 8, dtype=torch.bfloat16), torch.zeros(4, 8, dtype=torch.bfloat16))


def test_pil_bicubic_restatement_and_coefficients_match_pillow():
    """The restated Pillow 8-bit bicubic resample (oracle) and the product's coefficient tables against PIL itself."""
    from PIL import Image
    from oracle import ref_cpu as O
    from ufvideo_amd.mm_utils import pil_resize_coeffs
    rng = np.random.default_rng(3)
    for H, W, S in [(360, 640, 336), (224, 224, 336), (96, 54, 56), (35, 50, 56), (56, 56, 56)]:
        img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        ref = np.asarray(Image.fromarray(img).resize((S, S), resample=Image.BICUBIC))
        assert np.array_equal(O.pil_resize_bicubic_u8(img, S, S), ref), (H, W, S)
        for n_in in (H, W):
            b1, k1 = pil_resize_coeffs(n_in, S)
            b2, k2 = O.pil_resize_coeffs(n_in, S)
            assert np.array_equal(b1, b2) and np.array_equal(k1, k2)


def test_ann_to_mask_rle_and_photo_grid():
    """COCO RLE decode (no pycocotools in the image: checked with an independent encoder written here from the format
    description) and create_photo_grid / process_image(image_grid=True) shapes."""
    rng = np.random.default_rng(7)

    def encode_runs(mask):                                   # column-major runs, first run counts zeros
        flat = mask.T.reshape(-1)
        runs, cur, n = [], 0, 0
        for v in flat:
            if v == cur:
                n += 1
            else:
                runs.append(n); cur, n = v, 1
        runs.append(n)
        return runs

    def to_string(runs):                                     # pycocotools rleToString
        out = []
        for i, r in enumerate(runs):
            x = r - runs[i - 2] if i > 2 else r
            more = True
            while more:
                c = x & 0x1F
                x >>= 5
                more = not ((x == 0 and not (c & 0x10)) or (x == -1 and (c & 0x10)))
                if more:
                    c |= 0x20
                out.append(chr(c + 48))
        return "".join(out)

    for h, w in ((7, 5), (40, 33), (1, 9)):
        m = (rng.random((h, w)) > 0.6).astype(np.uint8)
        runs = encode_runs(m)
        assert np.array_equal(MU.annToMask({"counts": runs, "size": [h, w]}), m)
        assert np.array_equal(MU.annToMask({"counts": to_string(runs), "size": [h, w]}), m)
        assert np.array_equal(MU.annToMask({"counts": to_string(runs).encode(), "size": [h, w]}), m)
    with pytest.raises(NotImplementedError):
        MU.annToMask([[0, 0, 4, 0, 4, 4]], 8, 8)
    fr = rng.integers(0, 256, (5, 6, 4, 3), dtype=np.uint8)
    g = MU.create_photo_grid(fr)
    assert g.shape == (18, 8, 3) and np.array_equal(g[6:12, 4:8], fr[3]) and g[12:, 4:].sum() == 0
    from PIL import Image
    proc = MU.UfvImageProcessor(size=56)
    imgs, hh, ww, fl = MU.process_image(Image.fromarray(fr[0]), proc, num_frames=4, image_grid=True)
    assert imgs.shape == (2, 3, 56, 56) and (hh, ww) == (12, 8) and len(fl) == 4


def test_checkpoint_loading_reports_missing_and_unexpected_keys(tmp_path):
    """load_state_dict returns what nn.Module returns (minus the reference's unused SAM2-memory / tower-head tensors), and
    from_pretrained refuses a checkpoint that would leave weights at their random initialisation; tie_word_embeddings is honoured"""
    import json
    import torch
    from safetensors.torch import save_file
    from ufvideo_amd.model import VideoReferQwen2Config, VideoReferQwen2ForCausalLM
    kw = dict(vocab_size=64, hidden_size=32, intermediate_size=64, num_hidden_layers=1, num_attention_heads=2, num_key_value_heads=1,
              train_mask_decoder=True)
    m = VideoReferQwen2ForCausalLM(VideoReferQwen2Config(**kw))
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    full = dict(sd)
    full["model.mask_encoder.sam2_model.memory_encoder.fuser.layers.0.weight"] = torch.zeros(2)      # ignored by design
    full["something.else"] = torch.zeros(1)
    del full["model.layers.0.mlp.up_proj.weight"]
    res = m.load_state_dict(full, strict=False)
    assert res.missing_keys == ["model.layers.0.mlp.up_proj.weight"] and res.unexpected_keys == ["something.else"]
    with pytest.raises(KeyError):
        m.load_state_dict(full, strict=True)
    # from_pretrained: a missing tensor is an error, not a silent random init
    d = tmp_path / "ckpt"; d.mkdir()
    (d / "config.json").write_text(json.dumps(dict(kw, tie_word_embeddings=False)))
    part = {k: v.contiguous() for k, v in sd.items() if k != "model.norm.weight"}
    save_file(part, str(d / "model.safetensors"))
    with pytest.raises(KeyError, match="model.norm.weight"):
        VideoReferQwen2ForCausalLM.from_pretrained(str(d))
    # tied embeddings: the head is taken from embed_tokens
    tied = {k: v.contiguous() for k, v in sd.items() if k != "lm_head.weight"}
    save_file(tied, str(d / "model.safetensors"))
    (d / "config.json").write_text(json.dumps(dict(kw, tie_word_embeddings=True)))
    m2 = VideoReferQwen2ForCausalLM.from_pretrained(str(d))
    assert torch.equal(m2.lm_head.weight, m2.get_model().embed_tokens.weight) and torch.equal(m2.lm_head.weight, sd["model.embed_tokens.weight"])
    (d / "config.json").write_text(json.dumps(dict(kw, tie_word_embeddings=False)))
    with pytest.raises(KeyError, match="lm_head.weight"):
        VideoReferQwen2ForCausalLM.from_pretrained(str(d))


def test_packed_module_refuses_repacking_while_a_trainer_owns_it():
    from ufvideo_amd.model._params import PackedModule, set_gemm_dtype

    class M(PackedModule):
        def _pack(self):
            return {}
    m = M()
    m.packed(); m.invalidate(); m.to("cpu")
    m._owner = object()
    for f in (m.invalidate, lambda: m.to("cpu"), lambda: m.load_state_dict({}), lambda: set_gemm_dtype(m, "bf16")):
        with pytest.raises(RuntimeError, match="DecoderTrainer owns"):
            f()
    m._owner = None
    m.invalidate()


def _tiny_tokenizer_dir(path, n_vocab=40):
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast
    vocab = {f"t{i}": i for i in range(n_vocab - 3)}
    vocab.update({"<unk>": n_vocab - 3, "<eos>": n_vocab - 2, "<pad>": n_vocab - 1})
    tok = Tokenizer(models.WordLevel(vocab, unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.WhitespaceSplit()
    PreTrainedTokenizerFast(tokenizer_object=tok, unk_token="<unk>", eos_token="<eos>", pad_token="<pad>").save_pretrained(path)


def test_lora_and_base_model_loading_forms(tmp_path):
    """The reference's other two loading forms (ufvideo/model/__init__.py:82-125): (i) lora=True -- base model, initialize_MM_tokenizer,
    non_lora_trainables.bin (with peft's key prefixes), adapter merged as W + (alpha / r) B A, incl. rank / alpha patterns, rsLoRA and
    modules_to_save; (ii) base + mm_projector.bin.  peft is not in the image: the merge is checked against its definition."""
    import json
    import torch
    from safetensors.torch import save_file
    from ufvideo_amd.model import VideoReferQwen2Config, VideoReferQwen2ForCausalLM, load_pretrained_model
    from ufvideo_amd.model.lora import merge_lora
    llm = dict(vocab_size=40, hidden_size=32, intermediate_size=64, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=1)
    mmk = dict(mm_vision_tower="siglip", mm_vision_select_layer=-2, mm_vision_select_feature="patch", mm_projector_type="spatial_conv", mm_hidden_size=16,
               mm_region_encoder_type="pooling", image_aspect_ratio="square", num_frames=2, seg_token_id=39, sam2_trunk=None,
               vision_config=dict(hidden_size=16, intermediate_size=32, num_hidden_layers=2, num_attention_heads=2, image_size=28, patch_size=14))
    full = VideoReferQwen2ForCausalLM(VideoReferQwen2Config(**llm, **mmk, train_mask_decoder=True), seed=3)
    sd = {k: v.clone() for k, v in full.state_dict().items()}
    base, adir, pdir = tmp_path / "base", tmp_path / "adapter", tmp_path / "pretrain"
    for d in (base, adir, pdir):
        d.mkdir()
    # a plain language-model base: no multimodal modules
    lm_keys = [k for k in sd if not k.startswith(("model.mm_projector.", "model.region_encoder.", "model.text_hidden_fcs.", "model.vision_tower."))]
    save_file({k: sd[k].contiguous() for k in lm_keys}, str(base / "model.safetensors"))
    (base / "config.json").write_text(json.dumps(dict(llm, model_type="qwen2")))
    _tiny_tokenizer_dir(str(base))
    # ---- (i) the adapter directory
    (adir / "config.json").write_text(json.dumps(dict(llm, **mmk, model_type="videorefer_qwen2")))
    _tiny_tokenizer_dir(str(adir))
    g = torch.Generator().manual_seed(5)
    r, alpha = 4, 8
    targets = {"model.layers.0.self_attn.q_proj": (r, alpha), "model.layers.1.self_attn.v_proj": (r, alpha), "model.layers.1.mlp.down_proj": (2, 6)}
    asd = {}
    for mod, (rr, _) in targets.items():
        out_f, in_f = sd[mod + ".weight"].shape
        asd[f"base_model.model.{mod}.lora_A.weight"] = torch.randn(rr, in_f, generator=g) * 0.3
        asd[f"base_model.model.{mod}.lora_B.weight"] = torch.randn(out_f, rr, generator=g) * 0.3
    new_norm = torch.randn(32, generator=g)
    asd["base_model.model.model.norm.modules_to_save.weight"] = new_norm
    save_file({k: v.contiguous() for k, v in asd.items()}, str(adir / "adapter_model.safetensors"))
    acfg = dict(peft_type="LORA", r=r, lora_alpha=alpha, target_modules=["q_proj", "v_proj", "down_proj"], fan_in_fan_out=False,
                rank_pattern={"down_proj": 2}, alpha_pattern={"down_proj": 6}, modules_to_save=["norm"])
    (adir / "adapter_config.json").write_text(json.dumps(acfg))
    proj = {k: torch.randn(v.shape, generator=g) * 0.1 for k, v in sd.items() if k.startswith("model.mm_projector.")}
    torch.save({"base_model.model." + k: v for k, v in proj.items()}, str(adir / "non_lora_trainables.bin"))
    tok, m, processor, ctx = load_pretrained_model(str(adir), str(base), "videorefer_qwen2", device="cpu", lora=True)
    n_tok = len(tok)
    assert n_tok == 40 + 102 and m.lm_head.weight.shape[0] == n_tok and m.config.vocab_size == n_tok       # <region>, 100 x <TEMP-..>, [SEG]
    got = dict(m.state_dict())
    for mod, (rr, al) in targets.items():
        want = sd[mod + ".weight"].float() + (al / rr) * (asd[f"base_model.model.{mod}.lora_B.weight"] @ asd[f"base_model.model.{mod}.lora_A.weight"])
        assert torch.allclose(got[mod + ".weight"].float(), want.to(torch.bfloat16).float(), atol=1e-6), mod
        assert not torch.equal(got[mod + ".weight"], sd[mod + ".weight"])
    assert torch.equal(got["model.norm.weight"], new_norm.to(torch.bfloat16))
    assert torch.equal(got["model.layers.0.self_attn.k_proj.weight"], sd["model.layers.0.self_attn.k_proj.weight"])          # untouched
    for k, v in proj.items():
        assert torch.equal(got[k], v.to(torch.bfloat16)), k
    assert torch.equal(got["model.embed_tokens.weight"][:40], sd["model.embed_tokens.weight"])
    # the merge itself: rsLoRA scaling, fan_in_fan_out, refusals
    p = {"x.weight": torch.zeros(6, 5)}
    A, B = torch.randn(3, 5, generator=g), torch.randn(6, 3, generator=g)
    merge_lora(p, dict(r=3, lora_alpha=6, use_rslora=True), {"base_model.model.x.lora_A.default.weight": A, "base_model.model.x.lora_B.default.weight": B})
    assert torch.allclose(p["x.weight"], (6 / 3 ** 0.5) * (B @ A), atol=1e-6)
    p = {"x.weight": torch.zeros(5, 6)}
    merge_lora(p, dict(r=3, lora_alpha=3, fan_in_fan_out=True), {"x.lora_A.weight": A, "x.lora_B.weight": B})
    assert torch.allclose(p["x.weight"], (B @ A).t(), atol=1e-6)
    with pytest.raises(NotImplementedError):
        merge_lora(p, dict(r=3, lora_alpha=3, use_dora=True), {})
    with pytest.raises(KeyError, match="does not have"):
        merge_lora({}, dict(r=3, lora_alpha=3), {"y.lora_A.weight": A, "y.lora_B.weight": B})
    with pytest.raises(ValueError, match="model_base"):
        load_pretrained_model(str(adir), None, "videorefer_qwen2", device="cpu", lora=True)
    # ---- (ii) base + mm_projector.bin (the reference's pre-training form)
    (pdir / "config.json").write_text(json.dumps(dict(llm, **mmk, model_type="videorefer_qwen2", tune_mm_mlp_adapter=True)))
    torch.save(proj, str(pdir / "mm_projector.bin"))
    tok2, m2, _, _ = load_pretrained_model(str(pdir), str(base), "videorefer_qwen2", device="cpu")
    got2 = dict(m2.state_dict())
    for k, v in proj.items():
        assert torch.equal(got2[k], v.to(torch.float16).to(torch.bfloat16)), k
    assert torch.equal(got2["model.layers.1.mlp.down_proj.weight"], sd["model.layers.1.mlp.down_proj.weight"]) and len(tok2) == 40
    with pytest.raises(NotImplementedError):
        load_pretrained_model(str(pdir), str(base), "videorefer_qwen2", device="cpu", load_4bit=True)


def test_gemm_kernel_choice_at_the_config2_shapes():
    """The cost model behind UFV_GEMM_AUTO (csrc/gemm.hip choose_kernel) is host arithmetic: pin what it picks at the shapes of the headline clip
    (measured best or within a few % of it on MI355X: tools/gemm_shapes.py, tools/gemm_splitk.py) so that a change of its constants shows up here."""
    from ufvideo_amd import _lib
    pick = _lib.load().ufv_gemm_choice
    assert pick(2399, 37888, 3584, 0, 1, 1) == 1442                      # gate/up (SwiGLU): 256x256
    assert pick(18432, 3456, 1152, 0, 0, 1) == 1442 and pick(18432, 4352, 1152, 0, 0, 0) == 1442     # ViT qkv, fc1 (GELU)
    assert pick(18432, 1152, 1152, 1, 0, 1) == 1431 and pick(18432, 1152, 4352, 1, 0, 1) == 1431     # ViT out_proj, fc2: 224x192, 498 tiles = 1.95 rounds
    assert pick(2399, 3584, 3584, 1, 0, 1) == 1331                        # LLM o_proj: 192x192, 247 tiles = one round
    assert pick(2399, 4608, 3584, 0, 0, 1) in (1332, 1441)                # LLM qkv
    # LLM down: since the row-pipelined residual epilogue (round 3) the unsplit 192x192 kernel (247 tiles = one full round) beats the split-K form
    assert pick(2399, 3584, 18944, 1, 0, 1) == 1331 and pick(4703, 3584, 18944, 1, 0, 1) == 1331
    assert pick(1200, 3584, 18944, 1, 0, 1) // 10000 >= 4                 # a short prompt leaves CUs idle unsplit: aligned split-K
    assert pick(1200, 3584, 18944, 1, 0, 0) < 10000                       # ... never split under an activation
    assert pick(1200, 3584, 18944, 0, 0, 1) < 10000                       # ... or into a bf16 output
    assert pick(100, 3584, 3584, 1, 0, 1) == 0 and pick(2399, 3500, 3584, 1, 0, 1) == 0      # small / unaligned shapes: not the ping-pong kernel
    # the connector: stage-2 1x1 convolutions / readout and the Conv3d sampler as a GEMM (bf16 outputs, 2304 rows): 192x192, 228 tiles = one round
    assert pick(2304, 3584, 3584, 0, 0, 1) == 1331 and pick(2304, 3584, 28672, 0, 0, 1) == 1331
    assert pick(18432, 3584, 3584, 0, 0, 1) == 1442                       # stage-1 1x1 convolutions: 1008 tiles of 256x256


def test_generated_kernel_bodies_are_what_their_generators_emit(tmp_path):
    """csrc/attn_vit_p2_asm.inc and csrc/attn_c128_asm.inc are GENERATED files that travel with the source (the GPU box only compiles): the committed text must
    be exactly what tools/gen_attn_p2.py / tools/gen_attn_c128.py write with no options (no lab knobs, no stamps), so that nobody reviews one thing and ships another"""
    import shutil
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if not k.startswith("UFV_")}
    for gen, inc, opts in (("gen_attn_p2.py", "attn_vit_p2_asm.inc", []), ("gen_attn_p2.py", "attn_vit_p2_s729_asm.inc", ["--seq", "729"]), ("gen_attn_c128.py", "attn_c128_asm.inc", [])):
        work = tmp_path / (gen + "".join(opts))
        (work / "tools").mkdir(parents=True)
        (work / "ufvideo_amd" / "csrc").mkdir(parents=True)
        shutil.copy(os.path.join(root, "tools", gen), work / "tools" / gen)
        subprocess.run([sys.executable, str(work / "tools" / gen)] + opts, check=True, env=env, stdout=subprocess.DEVNULL)
        fresh = (work / "ufvideo_amd" / "csrc" / inc).read_text()
        assert fresh == open(os.path.join(root, "ufvideo_amd", "csrc", inc)).read(), f"{inc} is stale: run tools/{gen}"
