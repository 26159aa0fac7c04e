"""The C-ABI shared library loads on a GPU-less box and exports every symbol include/mart_hip.h declares
(no compute calls here).  Also: the product path refuses to run without a HIP device."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    from mkg_analogy_amd import _lib
    return _lib


def _declared():
    txt = open(os.path.join(ROOT, "include", "mart_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mart_[a-z0-9_]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported_and_bound(built):
    l = built.lib()
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(l, n), f"{n} declared in mart_hip.h but not exported by libmart_hip.so"
    assert set(names) == set(built.EXPORTS), set(names) ^ set(built.EXPORTS)
    assert l.mart_abi_version() == built.EXPECTED_ABI == 10


def test_descriptor_layouts_match_header(built):
    """ctypes mirrors must have the field order of the C structs (spot-check by name order in the header)."""
    txt = open(os.path.join(ROOT, "include", "mart_hip.h")).read()
    for cname, cls in (("mart_gemm_nt_desc", built.GemmNT), ("mart_gemm_tn_desc", built.GemmTN), ("mart_ln_fwd_desc", built.LnFwd),
                       ("mart_ln_bwd_desc", built.LnBwd), ("mart_text_embed_desc", built.TextEmbed), ("mart_attn_fwd_desc", built.AttnFwd),
                       ("mart_adamw_desc", built.AdamW), ("mart_fusion_fwd_desc", built.FusionFwd), ("mart_fusion_bwd_desc", built.FusionBwd)):
        end = txt.index("} " + cname + ";")
        body = txt[txt.rindex("typedef struct {", 0, end) + len("typedef struct {"):end]
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            # "const void* A" / "int lda, ldb" / "long long stride_a, stride_b"
            names = re.sub(r"^(const\s+)?(unsigned\s+)?(long long|[a-z0-9_]+_t|[a-z]+)\s*\**", "", decl)
            fields += [n.strip().lstrip("*").strip() for n in names.split(",")]
        got = [f[0] for f in cls._fields_]
        norm = lambda s: s.rstrip("_")
        assert [norm(f) for f in fields] == [norm(g) for g in got], (cname, fields, got)


def test_descriptor_sizes_match_a_c_compiler(built, tmp_path):
    """sizeof / last-field offset of every descriptor as gcc sees include/mart_hip.h == the ctypes mirrors
    (catches a field added on one side only, or a type width mismatch, without a GPU)."""
    import ctypes
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    pairs = [("mart_gemm_nt_desc", built.GemmNT), ("mart_gemm_tn_desc", built.GemmTN), ("mart_ln_fwd_desc", built.LnFwd),
             ("mart_ln_bwd_desc", built.LnBwd), ("mart_text_embed_desc", built.TextEmbed), ("mart_attn_fwd_desc", built.AttnFwd),
             ("mart_attn_bwd_desc", built.AttnBwd), ("mart_adamw_desc", built.AdamW), ("mart_attn_f32_desc", built.AttnF32),
             ("mart_fusion_fwd_desc", built.FusionFwd), ("mart_fusion_bwd_desc", built.FusionBwd)]
    src = tmp_path / "sz.c"
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "mart_hip.h"', 'int main(void) {']
    for cname, cls in pairs:
        last = cls._fields_[-1][0]
        lines.append(f'  printf("{cname} %zu %zu\\n", sizeof({cname}), offsetof({cname}, {last}));')
    lines += ['  return 0;', '}']
    src.write_text("\n".join(lines))
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split("\n")
    seen = {l.split()[0]: (int(l.split()[1]), int(l.split()[2])) for l in out if l.strip()}
    for cname, cls in pairs:
        last = cls._fields_[-1][0]
        assert seen[cname] == (ctypes.sizeof(cls), getattr(cls, last).offset), (cname, seen[cname], ctypes.sizeof(cls), getattr(cls, last).offset)


def test_integration_doc_stub_matches_binding(built):
    """The ctypes stub a maintainer would paste from INTEGRATION.md has the same fields as the shipped binding."""
    txt = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    body = txt[txt.index("class GemmNT(C.Structure)"):txt.index("lib.mart_gemm_nt.argtypes")]
    names = re.findall(r'\("([A-Za-z0-9_]+)",\s*C\.c_', body)
    assert names == [f[0] for f in built.GemmNT._fields_]


def test_product_fails_loudly_without_gpu(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from mkg_analogy_amd import ops
    with pytest.raises(built.MartError):
        ops.require_gpu()
    from mkg_analogy_amd.models import MKGformerKGC, TextConfig, VisionConfig
    tc = TextConfig(vocab_size=64, hidden_size=768, num_hidden_layers=12, intermediate_size=3072, max_position_embeddings=32)
    m = MKGformerKGC(VisionConfig(patch_size=32), tc)
    ids = torch.zeros(1, 8, dtype=torch.long)
    with pytest.raises(built.MartError):
        m(input_ids=ids, attention_mask=torch.ones_like(ids), token_type_ids=ids, pixel_values=torch.zeros(1, 2, 3, 224, 224), return_dict=True)
