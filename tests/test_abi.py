"""CPU tests of the drop-in boundary: the C-ABI library builds, loads and exports every symbol that
include/hallo_amd.h declares; argument validation works without touching a GPU."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from hallo_amd import build, lib as hl
    build.build()
    return hl.load()


def test_header_symbols_are_exported(lib):
    from hallo_amd import lib as hl
    hdr = open(os.path.join(ROOT, "include", "hallo_amd.h")).read()
    declared = set(re.findall(r"^(?:int|int64_t|const char\*)\s+(hallo_\w+)\s*\(", hdr, flags=re.M))
    assert declared, "no declarations parsed from include/hallo_amd.h"
    assert declared == set(hl.SYMBOLS), (declared ^ set(hl.SYMBOLS))
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.hallo_abi_version() == 9


def test_struct_sizes_match_header(lib):
    """The ctypes mirrors must have the layout a C compiler gives the header's structs."""
    import subprocess
    import tempfile
    from hallo_amd import lib as hl
    src = '#include "hallo_amd.h"\n#include <stdio.h>\nint main(){printf("%zu %zu %zu\\n", sizeof(hallo_gemm_desc),' \
          ' sizeof(hallo_conv_desc), sizeof(hallo_attn_desc));return 0;}\n'
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "s.c")
        open(c, "w").write(src)
        exe = os.path.join(td, "s")
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe], check=True)
        out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split()
    assert [int(v) for v in out] == [C.sizeof(hl.GemmDesc), C.sizeof(hl.ConvDesc), C.sizeof(hl.AttnDesc)]


def test_argument_validation_returns_einval(lib):
    from hallo_amd import lib as hl
    d = hl.GemmDesc()
    assert lib.hallo_gemm(C.byref(d), None) == -22          # null pointers
    a = hl.AttnDesc()
    assert lib.hallo_attention(C.byref(a), None) == -22
    c = hl.ConvDesc()
    assert lib.hallo_conv3x3_nhwc(C.byref(c), None) == -22
    assert lib.hallo_temporal_attention(None, None, 1, 18, 64, 320, 8, 0.1, 1, None) == -22
    assert lib.hallo_groupnorm_chunks(4096) == 64 and lib.hallo_groupnorm_chunks(4) == 1
    assert lib.hallo_set_option(b"no_such_option", 1) == -22 and lib.hallo_set_option(b"gemm_variant", 2) == 0
    assert lib.hallo_set_option(b"gemm_variant", 9) == -22 and lib.hallo_set_option(b"gemm_variant", 6) == 0   # back to auto
    assert lib.hallo_groupnorm_chunks(64 * 64 * 64) == 64


def test_missing_library_fails_loudly(tmp_path):
    from hallo_amd import lib as hl
    with pytest.raises(hl.HalloLibraryError):
        hl.load(str(tmp_path / "nope.so"))


def test_cpu_tensors_are_rejected():
    """There is no CPU fallback: operators refuse host tensors."""
    import torch
    from hallo_amd import ops, lib as hl
    a = torch.zeros(8, 8, dtype=torch.bfloat16)
    with pytest.raises(hl.HalloLibraryError):
        ops.gemm(a, a)

def test_every_entry_point_rejects_bad_arguments(lib):
    """Error behaviour of the C ABI (include/hallo_amd.h: status -22 = bad argument): every entry point validates its
    arguments BEFORE touching the device, so this runs without a GPU -- null descriptors / pointers, sizes that break the
    documented alignment rules, unknown dtype / activation / option names, head dims the kernels are not built for."""
    import ctypes as C
    from hallo_amd import lib as L
    N=None
    res={}
    d=L.GemmDesc(); res['gemm_null']=lib.hallo_gemm(None,N); res['gemm_zero']=lib.hallo_gemm(C.byref(d),N)
    buf=(C.c_char*4096)(); p=C.cast(buf,C.c_void_p)
    d=L.GemmDesc(); d.A=d.B=d.C=p.value; d.M=8; d.N=8; d.K=12; d.lda=d.ldb=d.ldc=16; d.batch=1
    res['gemm_K_not_mult8']=lib.hallo_gemm(C.byref(d),N)
    d.K=16; d.lda=12; res['gemm_lda']=lib.hallo_gemm(C.byref(d),N)
    d.lda=16; d.act=7; res['gemm_act']=lib.hallo_gemm(C.byref(d),N)
    d.act=0; d.dtype=5; res['gemm_dtype']=lib.hallo_gemm(C.byref(d),N)
    # ln_parts: one (sum, sum of squares) pair per 64-column block of K, at most 32 of them (the staging LDS of a tile)
    d=L.GemmDesc(); d.A=d.B=d.C=p.value; d.M=128; d.N=128; d.K=320; d.lda=d.ldb=320; d.ldc=128; d.batch=1
    d.ln_colsum=p.value; d.ln_stats=p.value; d.ln_eps=1e-5
    d.ln_parts=4; res['gemm_ln_parts_not_K_over_64']=lib.hallo_gemm(C.byref(d),N)
    d.K=2560; d.lda=d.ldb=2560; d.ln_parts=40; res['gemm_ln_parts_over_32']=lib.hallo_gemm(C.byref(d),N)
    res['temporal_lead_neg']=lib.hallo_temporal_attention_lead(p,p,2,18,-1,4,80,2,1.0,0,N)
    res['temporal_lead_all']=lib.hallo_temporal_attention_lead(p,p,2,18,18,4,80,2,1.0,0,N)
    c=L.ConvDesc(); res['conv_zero']=lib.hallo_conv3x3_nhwc(C.byref(c),N); res['conv_null']=lib.hallo_conv3x3_nhwc(None,N)
    a=L.AttnDesc(); res['attn_zero']=lib.hallo_attention(C.byref(a),N); res['attn_null']=lib.hallo_attention(None,N)
    a=L.AttnDesc(); a.q=a.k1=a.v1=a.o=p.value; a.batch=1; a.heads=2; a.head_dim=64; a.Lq=8; a.Lkv1=8
    res['attn_hd64']=lib.hallo_attention(C.byref(a),N)
    res['temporal_null']=lib.hallo_temporal_attention(N,N,1,4,4,80,2,1.0,0,N)
    res['temporal_F33']=lib.hallo_temporal_attention(p,p,1,33,4,80,2,1.0,0,N)
    res['gn_null']=lib.hallo_groupnorm_nhwc(N,N,N,N,N,1,4,32,4,1e-5,0,0,N)
    res['gn_cpg1']=lib.hallo_groupnorm_nhwc(p,p,p,p,p,1,4,32,32,1e-5,0,0,N)
    res['gn2_C1_not8']=lib.hallo_groupnorm_nhwc2(p,20,p,p,p,p,p,1,4,64,4,1e-5,0,0,N)
    res['gn2_no_x2']=lib.hallo_groupnorm_nhwc2(p,32,N,p,p,p,p,1,4,64,4,1e-5,0,0,N)
    res['ln_null']=lib.hallo_layernorm(N,N,N,N,N,4,32,1e-5,1,1,0,N)
    res['ln_C_not8']=lib.hallo_layernorm(p,p,p,p,N,4,20,1e-5,1,1,0,N)
    res['softmax_null']=lib.hallo_softmax_rows(N,N,4,8,1.0,0,N)
    res['softmax_cols']=lib.hallo_softmax_rows(p,p,4,6,1.0,0,N)
    res['copy2d_null']=lib.hallo_copy2d(N,8,N,8,4,8,0,N)
    res['copy2d_w']=lib.hallo_copy2d(p,8,p,8,4,6,0,N)
    res['nchw_null']=lib.hallo_nchw_to_nhwc(N,N,1,4,16,8,0,0,N)
    res['nhwc_null']=lib.hallo_nhwc_to_nchw_f32(N,N,1,4,16,8,1.0,0.0,0.0,1.0,0,N)
    res['rowstats_null']=lib.hallo_row_stats(N,N,4,320,1e-5,0,N)
    res['face_null']=lib.hallo_face_xattn(N,N,N,N,N,N,N,32,320,32,1e-5,0,N)
    res['face_C']=lib.hallo_face_xattn(p,p,p,p,p,p,p,32,40,32,1e-5,0,N)
    res['u8_null']=lib.hallo_frames_to_uint8(N,N,1,3,16,N)
    res['temb_null']=lib.hallo_timestep_embedding(N,N,1,320,0,N)
    res['ddim_null']=lib.hallo_cfg_ddim_step(N,8,N,N,8,4,4,0,1.0,0.5,0.6,0,N)
    res['ddim_alpha']=lib.hallo_cfg_ddim_step(p,8,p,p,8,4,4,0,1.0,1.5,0.6,0,N)
    res['w2v_ws_bad']=lib.hallo_w2v_conv0_workspace(4,512,10,5)
    res['w2v_null']=lib.hallo_w2v_conv0_gn_gelu(N,16000,N,N,N,N,N,512,10,5,1e-5,0,N)
    res['w2v_k']=lib.hallo_w2v_conv0_gn_gelu(p,16000,p,p,p,p,p,512,17,5,1e-5,0,N)
    res['lerp_null']=lib.hallo_lerp_rows(N,N,4,8,512,0,N)
    res['lerp_C']=lib.hallo_lerp_rows(p,p,4,8,20,0,N)
    res['ff_null']=lib.hallo_ff320(N,320,N,320,N,320,N,N,128,1,1e-5,0,N)
    res['ff_ld']=lib.hallo_ff320(p,324,p,320,p,320,p,p,128,1,1e-5,0,N)          # ldx not a multiple of 8
    res['ff_dtype']=lib.hallo_ff320(p,320,p,320,p,320,p,p,128,1,1e-5,7,N)
    res['ff_opt']=lib.hallo_set_option(b"ff_fused",10)
    res['rsdbg_gated']=lib.hallo_set_option(b"gemm_rs_dbg",1)                    # timing ablations need a -DHALLO_ABLATIONS build
    res['opt_unknown']=lib.hallo_set_option(b"nope",1); res['get_unknown']=lib.hallo_get_option(b"nope")
    bad = {k: v for k, v in res.items() if v != -22}
    assert not bad, bad
    assert len(res) >= 36
    assert lib.hallo_groupnorm_chunks(4096) > 0
    assert lib.hallo_ff320_pack_bytes() == 80 * 32768 and lib.hallo_get_option(b"gemm_rs_dbg") == 0 and lib.hallo_get_option(b"ff_fused") == 0
