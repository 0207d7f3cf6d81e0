"""CPU tests of the host side: drop-in surface (state-dict key names / shapes identical to the reference
architecture as restated by the oracle), plugin surface, error behaviour, no-fallback guarantees."""
import pytest
import torch

from oracle import hallo_ref as H
from oracle import harness as Hn


def _keys(m):
    return {k: tuple(v.shape) for k, v in m.state_dict().items()}


def test_state_dict_contract_full_size():
    """Full SD-1.5 widths on the meta device: 1946 / 682 entries with identical names+shapes."""
    from hallo_amd.models.unet_2d_condition import UNet2DConditionModel
    from hallo_amd.models.unet_3d import UNet3DConditionModel
    with torch.device("meta"):
        n3, o3 = UNet3DConditionModel(), H.UNet3DConditionModel()
        n2, o2 = UNet2DConditionModel(), H.UNet2DConditionModel()
    assert _keys(n3) == _keys(o3) and len(_keys(n3)) == 1946
    assert _keys(n2) == _keys(o2) and len(_keys(n2)) == 682
    assert sum(p.numel() for p in n3.parameters()) == sum(p.numel() for p in o3.parameters())


def test_state_dict_contract_small_modules():
    from diffusers import AutoencoderKL as OVAE
    from hallo_amd.models.audio_proj import AudioProjModel
    from hallo_amd.models.face_locator import FaceLocator
    from hallo_amd.models.image_proj import ImageProjModel
    from hallo_amd.models.vae import AutoencoderKL
    with torch.device("meta"):
        pairs = [(AutoencoderKL(), OVAE()), (FaceLocator(320), H.FaceLocator(320)),
                 (ImageProjModel(768, 512, 4), H.ImageProjModel(768, 512, 4)), (AudioProjModel(), H.AudioProjModel())]
    for a, b in pairs:
        assert _keys(a) == _keys(b)


def test_plugin_surface():
    from hallo_amd.models.unet_3d import HalloHipAttnProcessor, UNet3DConditionModel
    with torch.device("meta"):
        m = UNet3DConditionModel(**Hn.SMALL, audio_attention_dim=48, motion_module_kwargs=Hn.SMALL_MM)
    procs = m.attn_processors
    # 16 spatial blocks x (attn1, attn2) + 16 audio blocks x (attn1 + 3 cross) ; temporal transformers excluded
    assert len(procs) == 16 * 2 + 16 * 4 and all("temporal_transformer" not in k for k in procs)
    m.set_attn_processor(HalloHipAttnProcessor())
    from hallo_amd.attn_processor import HalloAttnProcessor          # the callable form of the same kernels
    m.set_attn_processor(HalloAttnProcessor())
    m.set_attn_processor({k: HalloAttnProcessor() for k in procs})
    m.set_attention_slice("auto")
    with pytest.raises(ValueError):
        m.set_attn_processor(object())            # no PyTorch / xformers fallback on this path
    with pytest.raises(ValueError):
        m.set_attn_processor({"x": HalloHipAttnProcessor()})
    assert m.enable_gradient_checkpointing() is None and m.config.cross_attention_dim == 64 and m.in_channels == 4


def test_unsupported_configurations_raise():
    from hallo_amd.models.unet_3d import UNet3DConditionModel
    with pytest.raises(ValueError):
        UNet3DConditionModel(use_motion_module=False)
    with pytest.raises(ValueError):
        UNet3DConditionModel(down_block_types=("DownBlock3D",) * 4)


def test_no_cpu_fallback():
    """CPU tensors / missing library fail loudly instead of silently computing somewhere else."""
    from hallo_amd import lib, ops
    from hallo_amd.models.image_proj import ImageProjModel
    with pytest.raises(lib.HalloLibraryError):
        ops.gemm(torch.zeros(8, 8, dtype=torch.float16), torch.zeros(8, 8, dtype=torch.float16))
    m = ImageProjModel(64, 512, 4)
    with pytest.raises(lib.HalloLibraryError):
        m.prepare()
    with pytest.raises(lib.HalloLibraryError):
        lib.load("/nonexistent/libhallo_amd.so")
    with pytest.raises(TypeError):
        ops.dtype_code(torch.float32)


def test_reference_control_contract():
    from hallo_amd.models.mutual_self_attention import ReferenceAttentionControl
    from hallo_amd.models.unet_2d_condition import UNet2DConditionModel
    from hallo_amd.models.unet_3d import UNet3DConditionModel
    with torch.device("meta"):
        den = UNet3DConditionModel(**Hn.SMALL, audio_attention_dim=48, motion_module_kwargs=Hn.SMALL_MM)
        ref = UNet2DConditionModel(**Hn.SMALL)
    w = ReferenceAttentionControl(ref, mode="write", fusion_blocks="full")
    r = ReferenceAttentionControl(den, mode="read", do_classifier_free_guidance=True, fusion_blocks="full")
    assert den.reference_do_cfg is True
    with pytest.raises(RuntimeError):
        r.update(w)                               # writer has not run
    ref.written_banks = [torch.ones(6, 4, 8) * 1.0001 for _ in range(16)]
    r.update(w)
    assert len(den.reference_bank) == 16 and den.reference_bank[0].dtype == torch.float16   # F4: always fp16
    r.clear()
    w.clear()
    assert den.reference_bank is None and ref.written_banks == []
    with pytest.raises(AssertionError):
        ReferenceAttentionControl(den, mode="bogus")


def test_clip_cache_refreshes_constants_in_place():
    """Graph mode (FaceAnimatePipeline(use_graph=True)): after begin_clip() the first get() of a key recomputes the value and
    copies it INTO the tensors of the previous clip, so a captured hipGraph keeps reading valid addresses; shape changes
    are refused."""
    import pytest
    import torch
    from hallo_amd.models.attention import ClipCache
    c = ClipCache()
    owner = object()
    made = []

    def make(val):
        def f():
            made.append(val)
            base = torch.full((2, 6), float(val))
            return base[:, :3], base[:, 3:]          # views of one buffer, like the fused k|v projections
        return f
    k0, v0 = c.get(owner, "kv", make(1))
    assert c.get(owner, "kv", make(2))[0] is k0 and made == [1]          # a hit: nothing recomputed
    c.begin_clip()
    k1, v1 = c.get(owner, "kv", make(3))
    assert k1 is k0 and v1 is v0 and made == [1, 3]
    assert float(k0[0, 0]) == 3.0 and float(v0[1, 2]) == 3.0             # refreshed inside the old storage
    assert c.get(owner, "kv", make(4))[0] is k0 and made == [1, 3]       # fresh again until the next clip
    c.begin_clip()
    with pytest.raises(ValueError):
        c.get(owner, "kv", lambda: (torch.zeros(2, 4), torch.zeros(2, 3)))
    c.clear()
    assert c.get(owner, "kv", make(5))[0] is not k0


def test_gemm4_schedule_covers_every_tile_and_k_step_once():
    """The work deal of csrc/gemm4.hip (hallo_gemm4_schedule, computed on the host: no GPU needed), replayed with the kernel's own
    segment rules for every workgroup: each (tile, K step) of the problem must be owned exactly once, the parts of a split tail
    tile must be contiguous in K and in workgroup index, and the XCD remap must be a bijection."""
    import ctypes as C
    from hallo_amd import build, lib as hl
    build.build()
    lib = hl.load()

    def xcd_remap(bid, nwg):
        q, r = nwg >> 3, nwg & 7
        xcd, idx = bid & 7, bid >> 3
        return (xcd * (q + 1) if xcd < r else r * (q + 1) + (xcd - r) * q) + idx

    ws = 128 << 20
    shapes = [(4096, 1280, 1280), (4096, 1280, 5120), (4096, 3840, 1280), (4608, 3840, 1280), (4608, 1280, 1280), (16384, 640, 640),
              (18432, 1920, 640), (1024, 1280, 5120), (1152, 3840, 1280), (1000, 1288, 1280), (5000, 648, 704), (128, 160, 256),
              (65536, 320, 1280), (130, 170, 4096), (256 * 128, 160, 640), (257 * 128, 160, 8192)]
    seen_split = False
    for (M, N, K) in shapes:
        sched = (C.c_int * 8)()
        ok = lib.hallo_gemm4_schedule(M, N, K, ws, 0, sched)
        assert ok == 1, (M, N, K, ok)
        tm, tn, nk, G, dp, R, parts, per = list(sched)
        assert tm == -(-M // 128) and tn == -(-N // 160) and nk == K // 64 and dp * G + R == tm * tn and 0 <= R < max(G, R + 1)
        assert sorted(xcd_remap(b, G) for b in range(G)) == list(range(G))
        if dp:
            assert sorted(xcd_remap(b, dp * G) for b in range(dp * G)) == list(range(dp * G))
        owned = {}
        for bid in range(G):
            wl = xcd_remap(bid, G)
            for j in range(dp):
                t = xcd_remap(j * G + bid, dp * G)
                for k in range(nk):
                    owned[(t, k)] = owned.get((t, k), 0) + 1
            if wl < R * parts:
                t, q = dp * G + wl // parts, wl % parts
                kb, ke = q * per, min(nk, q * per + per)
                assert kb < ke, (M, N, K, wl)                  # every part owns at least one K step
                for k in range(kb, ke):
                    owned[(t, k)] = owned.get((t, k), 0) + 1
        assert len(owned) == tm * tn * nk and set(owned.values()) == {1}, (M, N, K)
        if parts > 1:
            seen_split = True
            assert R * parts <= G and (parts - 1) * per < nk <= parts * per
    assert seen_split
    assert lib.hallo_gemm4_schedule(4096, 1280, 1000, ws, 0, (C.c_int * 8)()) == 0      # K % 64 != 0: not covered
    assert lib.hallo_gemm4_schedule(0, 1280, 1280, ws, 0, (C.c_int * 8)()) == -22


def test_routing_is_a_scope_not_process_state():
    """Round 5: the kernel routing belongs to a pipeline and is applied around its enqueue calls.  `ops.routing`
    sets only the options that differ, restores the previous values on exit (whatever they were -- not hard-coded defaults),
    and does not bump the option epoch when nothing changes; `options_fingerprint` is what captured graphs are keyed on."""
    from hallo_amd import ops
    base = ops.options_fingerprint()
    user_prev = ops.set_option("split_k_max", 8)                  # a user-set value that set_mode(False) used to clobber
    try:
        before = ops.options_fingerprint()
        e0 = ops.option_epoch()
        with ops.routing("throughput"):
            inside = ops.options_fingerprint()
            assert ops.get_option("gemm_rs") == 0 and ops.get_option("gn_fused") == 0 and ops.get_option("split_k_max") == 4
            e1 = ops.option_epoch()
            with ops.routing(dict(ops.THROUGHPUT_OPTIONS)):     # nested, same values: nothing is touched
                assert ops.option_epoch() == e1
            assert ops.options_fingerprint() == inside
        assert ops.options_fingerprint() == before and ops.get_option("split_k_max") == 8
        assert ops.option_epoch() > e0
        with ops.routing(None):
            assert ops.options_fingerprint() == before
        e2 = ops.option_epoch()
        assert ops.set_option("split_k_max", 8) == 8 and ops.option_epoch() == e2          # unchanged value: no epoch bump
    finally:
        ops.set_option("split_k_max", user_prev)
    assert ops.options_fingerprint() == base
    assert inside != before and len(base) == len(ops.routing_option_names()) >= 21 and all(v >= 0 for v in base)


def test_rank_core_slices_respect_the_cgroup_quota():
    """bench.py pins each rank to min(affinity share, quota share) cores (the pool's boxes show 256 cores in the mask
    under a 16-core cgroup quota)."""
    import bench
    cores = list(range(256))
    sl = [bench.rank_cores(cores, 16, r, 8) for r in range(8)]
    assert all(len(s) == 2 for s in sl) and sl[0] == [0, 1] and sl[7] == [224, 225]
    assert len({c for s in sl for c in s}) == 16
    assert bench.rank_cores(cores, None, 3, 8) == list(range(96, 128))             # no quota: the whole share of the mask
    assert bench.rank_cores(cores, 4, 1, 8) == [32]                                # quota below one core per rank: one core each
    assert bench.rank_cores(list(range(4)), 16, 0, 8) is None                      # fewer cores than ranks: no pinning
    assert bench.rank_cores(list(range(16)), 64, 1, 2) == list(range(8, 16))


def test_scratch_scope_selects_the_pipelines_own_scratch():
    """Round 5: launches enqueued inside `ops.scratch_scope(s)` use `s` whatever the current stream is (so a
    pipeline's eager launches and its graph captures -- which run on torch's process-wide capture stream -- share ITS buffers and
    nobody else's); scopes nest and unwind; a None scope falls through to the enclosing one."""
    from hallo_amd import ops

    class Fake:            # stands in for ops.Scratch (which allocates device memory): the selection logic is host-side
        def __init__(self, tag):
            self.splitk, self.tag = tag, tag
    a, b = Fake("a"), Fake("b")
    with ops.scratch_scope(a):
        assert ops.current_scratch("cpu") is a and ops._workspace("cpu") == "a"
        with ops.scratch_scope(b):
            assert ops.current_scratch("cpu") is b
            with ops.scratch_scope(None):
                assert ops.current_scratch("cpu") is b
            assert ops.current_scratch("cpu") is b
        assert ops.current_scratch("cpu") is a
    assert not ops._scratch_tls.stack
    try:
        with ops.scratch_scope(a):
            raise RuntimeError("x")
    except RuntimeError:
        pass
    assert not ops._scratch_tls.stack      # unwound on exceptions too


def test_scratch_scope_and_routing_are_safe_across_host_threads():
    """two host threads driving two pipelines.  The scratch scope stack is per thread (a thread never launches
    with another pipeline's split-K slab / GroupNorm scratch, and never pops the other's entry); the kernel routing is process
    state inside the library, so a routing scope holds a process-wide re-entrant lock: the second thread's scope starts when the
    first one's has ended, each sees ITS option values for the whole of its scope."""
    import threading
    import time
    from hallo_amd import ops

    class Fake:
        def __init__(self, tag):
            self.splitk = tag
    seen, errs = {}, []
    gate = threading.Barrier(2)
    base = ops.options_fingerprint()

    def work(tag, opts, hold):
        try:
            gate.wait()
            with ops.routing(opts), ops.scratch_scope(Fake(tag)):
                for _ in range(5):
                    assert ops._workspace("cpu") == tag                       # never the other thread's scratch
                    assert ops.get_option("split_k_max") == opts["split_k_max"]   # never the other thread's routing
                    time.sleep(hold)
                with ops.routing(opts):                                       # re-entrant for the owner
                    assert ops.get_option("gemm_rs") == opts["gemm_rs"]
                seen[tag] = ops.options_fingerprint()
        except Exception as e:      # noqa: BLE001
            errs.append((tag, repr(e)))
    ta = threading.Thread(target=work, args=("a", dict(ops.THROUGHPUT_OPTIONS, split_k_max=3), 0.01))
    tb = threading.Thread(target=work, args=("b", dict(ops.LATENCY_OPTIONS, split_k_max=7), 0.01))
    ta.start(); tb.start(); ta.join(); tb.join()
    assert not errs, errs
    assert seen["a"] != seen["b"] and ops.options_fingerprint() == base and not ops._scratch_tls.stack


def test_option_names_export_is_exhaustive(lib=None):
    """hallo_option_names (ABI v8): every listed option reads back through hallo_get_option and accepts its own value; the graph key
    of FaceAnimatePipeline is built from this list (a hand-kept tuple had missed 'xattn_cap')."""
    from hallo_amd import lib as L, ops
    lib = L.load()
    names = ops.routing_option_names()
    assert len(names) == len(set(names)) >= 21 and {"xattn_cap", "gemm_rs_dbg", "splitk_nt", "split_k_max", "gn_fused"} <= set(names)
    for n in names:
        v = lib.hallo_get_option(n.encode())
        assert v >= 0, n
        assert lib.hallo_set_option(n.encode(), v) == 0, n
    # and no settable option is missing from the list: the names in the sources' strcmp chains
    import os
    import re
    src = os.path.join(os.path.dirname(L.__file__), "csrc")
    found = set()
    for f in os.listdir(src):
        if f.endswith(".hip"):
            txt = open(os.path.join(src, f)).read()
            for m in re.finditer(r'strcmp\(name, "([a-z0-9_]+)"\)\)\s*(?:\|\|[^)]*\)\))?\s*\{\s*if \(value', txt):
                found.add(m.group(1))
    assert found and found <= set(names), found - set(names)


def test_kv_split_query_is_a_host_function():
    """hallo_gemm_kv_split_ok (ABI v9) answers without a GPU: the head-major K / V epilogue exists for the LayerNorm-fused q|k|v
    projection of the 320-channel level on the row-stationary kernel only -- K = 320, 640 K / V columns behind the q columns, at
    least 8192 rows, and not with the row-stationary kernels routed off."""
    from hallo_amd import lib
    l = lib.load()
    assert l.hallo_gemm_kv_split_ok(262144, 960, 320, 320) == 1
    assert l.hallo_gemm_kv_split_ok(65536, 960, 320, 320) == 1
    assert l.hallo_gemm_kv_split_ok(4096, 960, 320, 320) == 0          # below the kernel's minimum size
    assert l.hallo_gemm_kv_split_ok(65536, 1920, 640, 640) == 0        # the 640-channel level: another kernel
    assert l.hallo_gemm_kv_split_ok(65536, 960, 320, 0) == 0           # 960 columns are not K and V of 8 heads x 40
    assert l.hallo_gemm_kv_split_ok(65536, 640, 320, 0) == 1           # K | V alone (no q columns)
    old = l.hallo_get_option(b"gemm_rs")
    try:
        assert l.hallo_set_option(b"gemm_rs", 0) == 0
        assert l.hallo_gemm_kv_split_ok(262144, 960, 320, 320) == 0
    finally:
        l.hallo_set_option(b"gemm_rs", old)
