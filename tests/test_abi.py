"""The C-ABI library loads and exports every symbol include/moondream_hip.h declares."""
import ctypes
import os
import re

import pytest

from moondream_amd import _lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(REPO, "include", "moondream_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(md_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert declared_symbols() == _lib.exported_symbols()


def test_library_exports_every_declared_symbol():
    _lib.build_library(verbose=False)
    lib = _lib.load()
    for name in declared_symbols():
        assert hasattr(lib, name), name
    assert lib.md_abi_version() == _lib.ABI_VERSION == 5
    assert lib.md_status_string(0) == b"ok"
    assert b"workspace" in lib.md_status_string(3)


def test_argument_validation_needs_no_gpu():
    """Contract violations are rejected on the host, before any launch."""
    lib = _lib.load()
    args = _lib.MdGemmArgs()  # all null
    assert lib.md_gemm_bf16(ctypes.byref(args), None) == 1
    assert lib.md_gemm_bf16(None, None) == 1
    assert lib.md_vit_workspace_bytes(None, 4) == 0


def test_fp8_entry_points_validate_on_the_host():
    lib = _lib.load()
    assert lib.md_gemm_f8(None, None) == 1
    assert lib.md_gemm_f8(ctypes.byref(_lib.MdGemmF8Args()), None) == 1
    assert lib.md_quantize_f8(None, 0, None, 0, 1, 8, 8, 1.0, None) == 1
    assert lib.md_amax_bf16(None, 0, 1, 8, None, None) == 1
    assert lib.md_decode_step_b1_supported(None, None) == 0


def test_product_never_imports_the_oracle():
    pkg = os.path.join(REPO, "moondream_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "moondream_oracle" not in src and "oracle/" not in src, f


def test_graft_entry_build_loads_the_library():
    """The driver's build check: __graft_entry__.build() must succeed on a CPU-only box (cross-compile, dlopen, ABI
    version).  (It once asserted a stale ABI number.)"""
    import __graft_entry__ as g

    g.build()


def test_struct_layouts_match_the_header(tmp_path):
    """Every argument struct of include/moondream_hip.h against its ctypes mirror: total size and the offset of every
    field, as gcc lays the header out (a field added to one side only would shift everything behind it silently)."""
    import ctypes as C
    import shutil
    import subprocess

    from moondream_amd import _lib

    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    pairs = {
        "md_linear": _lib.MdLinear, "md_layernorm": _lib.MdLayerNorm, "md_gemm_args": _lib.MdGemmArgs, "md_linear_fp8": _lib.MdLinearFp8,
        "md_linear_f8": _lib.MdLinearF8, "md_gemm_f8_args": _lib.MdGemmF8Args, "md_attn_args": _lib.MdAttnArgs,
        "md_vit_block": _lib.MdVitBlock, "md_vit_block_f8": _lib.MdVitBlockF8, "md_vit_f8": _lib.MdVitF8, "md_vit_model": _lib.MdVitModel,
        "md_text_block": _lib.MdTextBlock, "md_text_block_fp8": _lib.MdTextBlockFp8, "md_text_fp8": _lib.MdTextFp8,
        "md_text_block_f8": _lib.MdTextBlockF8, "md_text_f8": _lib.MdTextF8, "md_text_model": _lib.MdTextModel,
        "md_kv_cache": _lib.MdKvCache, "md_lora_pair": _lib.MdLoraPair, "md_text_block_lora": _lib.MdTextBlockLora,
    }
    header = os.path.join(REPO, "include", "moondream_hip.h")
    text = open(header).read()
    # field names per struct from the header text (declarations between "typedef struct {" and "} name;")
    import re

    lines = ["#include <stdio.h>", "#include <stddef.h>", f'#include "{header}"', "int main(void) {"]
    fields = {}
    for name in pairs:
        m = re.search(r"typedef struct \{((?:(?!typedef struct).)*?)\} " + name + ";", text, flags=re.S)
        assert m, name
        body = re.sub(r"/\*.*?\*/", "", m.group(1), flags=re.S)
        names = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            # "type a, b, c" / "const type* a" / "type a[N]"
            first, *rest = decl.split(",")
            names.append(re.sub(r"\[.*?\]", "", first.split()[-1]).lstrip("*"))
            names += [re.sub(r"\[.*?\]", "", r.strip()).lstrip("*") for r in rest]
        fields[name] = names
        lines.append(f'  printf("{name} %zu", sizeof({name}));')
        for f in names:
            lines.append(f'  printf(" %zu", offsetof({name}, {f}));')
        lines.append('  printf("\\n");')
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c11", "-o", str(exe), str(src)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.strip().splitlines()
    assert len(out) == len(pairs)
    for line in out:
        name, size, *offs = line.split()
        cls = pairs[name]
        assert C.sizeof(cls) == int(size), (name, C.sizeof(cls), size)
        cnames = [f[0] for f in cls._fields_]
        assert cnames == fields[name], (name, cnames, fields[name])
        assert [getattr(cls, f).offset for f in cnames] == [int(o) for o in offs], name
