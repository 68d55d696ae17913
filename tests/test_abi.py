"""The C-ABI library loads on a CPU-only box and exports every symbol include/filtlong_hip.h declares
(no compute calls here: there is no GPU and no CPU fallback)."""
import ctypes
import os
import re

import pytest

from filtlong_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "filtlong_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(flx_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return ctypes.CDLL(_lib.LIB_PATH)


def test_header_and_loader_agree():
    assert header_functions() == sorted(_lib.ABI_SYMBOLS)


def test_library_exports_every_declared_symbol(lib):
    for name in header_functions():
        assert hasattr(lib, name), "missing export: " + name


def test_version_and_layout_helpers(lib):
    L = _lib.load()
    assert L.flx_abi_version() == 4
    assert b"gfx950" in L.flx_version()
    import numpy as np
    from filtlong_amd import api
    plane, offsets, lengths = api.pack_reads([b"ABC", b"", b"x" * 16, b"y" * 17])
    assert list(offsets) == [0, 16, 16, 32] and plane.nbytes == 64 and list(lengths) == [3, 0, 16, 17]
    assert bytes(plane[:3]) == b"ABC" and bytes(plane[32:49]) == b"y" * 17
    assert list(api.length_order(np.array([5, 9, 9, 1], dtype=np.int32))) == [1, 2, 0, 3]


def test_no_gpu_fails_loudly():
    """Without a gfx950 device the context must refuse (no silent CPU fallback)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from filtlong_amd import api
    with pytest.raises(api.FlxError):
        api.Context(0)


def _getenv_names(paths):
    names = set()
    for p in paths:
        names |= set(re.findall(r'getenv\("(FLX_[A-Z0-9_]+)"\)', open(p).read()))
    return names


def test_environment_switch_lists_match_the_sources(lib):
    """Round-4 review, item 9: an unknown name under one of the library's own prefixes is refused (flx_ctx_create for FLX_KMER_ /
    FLX_PHRED_ / FLX_RANK_ / FLX_RCCL_ / FLX_API_, the command line for FLX_CLI_*) instead of being ignored; any other FLX_ name is left alone.  The two lists must hold exactly the names the sources read, README.md must name every one
    of them, and the check itself must fire — before any device is asked for, so it runs without a GPU."""
    import glob
    import subprocess
    import sys
    csrc = glob.glob(os.path.join(ROOT, "filtlong_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "filtlong_amd", "csrc", "*.h"))
    cli = glob.glob(os.path.join(ROOT, "filtlong_amd", "cli", "*.h")) + glob.glob(os.path.join(ROOT, "filtlong_amd", "cli", "*.cpp"))
    ctx_src = open(os.path.join(ROOT, "filtlong_amd", "csrc", "flx_ctx.hip")).read()
    known_lib = set(re.findall(r'"(FLX_[A-Z0-9_]+)"', ctx_src[ctx_src.index("kKnownEnv[] = {"):ctx_src.index("static const char *const kOwnPrefixes[]")]))
    hosts = {"FLX_DEVICE", "FLX_COMM_ID_FILE", "FLX_LIB_PATH", "FLX_NO_TORCH_PRELOAD"}
    assert _getenv_names(csrc) == known_lib
    own = set(re.findall(r'"(FLX_[A-Z]+_)"', ctx_src[ctx_src.index("kOwnPrefixes[] = {"):ctx_src.index("extern char **environ;")]))
    assert own == {"FLX_KMER_", "FLX_PHRED_", "FLX_RANK_", "FLX_RCCL_", "FLX_API_"}
    assert all(any(n.startswith(p) for p in own) for n in known_lib)  # every switch the library reads lies under a prefix it checks
    main_src = open(os.path.join(ROOT, "filtlong_amd", "cli", "main.cpp")).read()
    known_cli = set(re.findall(r'"(FLX_CLI_[A-Z0-9_]+)"', main_src[main_src.index("static int check_cli_environment()"):main_src.index("int main(int argc")]))
    read_cli = _getenv_names(cli)
    assert {n for n in read_cli if n.startswith("FLX_CLI_")} == known_cli
    assert {n for n in read_cli if not n.startswith("FLX_CLI_")} <= hosts
    readme = open(os.path.join(ROOT, "README.md")).read()
    for name in sorted(known_lib | known_cli):
        assert name in readme, name + " is not in README.md"
    # the checks fire
    code = "from filtlong_amd import api\ntry:\n    api.Context(0)\nexcept api.FlxError as e:\n    print('REFUSED', e)\n"
    env = dict(os.environ, FLX_KMER_COVR="v2", PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT).stdout.decode()
    assert "REFUSED" in out and "FLX_KMER_COVR" in out, out
    # ... and a FOREIGN name (another tool's variable, a stale export) is none of the library's business: the drop-in binary must run
    # where the reference would (round-5 review, item 5) — without a GPU the context then fails for the device, not for the name
    env = dict(os.environ, FLX_FOO="1", FLX_SOMETHING_ELSE="x", PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, "-c", code.replace("api.Context(0)", "api.Context(0); print('CREATED')")], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT).stdout.decode()
    assert "unknown environment variable" not in out and ("CREATED" in out or "no HIP device" in out), out
    exe = os.path.join(ROOT, "filtlong_amd", "bin", "filtlong")
    if os.path.exists(exe):
        fq = os.path.join(ROOT, "tests", "golden", "ref_fixtures", "test_sort.fastq")
        p = subprocess.run([exe, "--target_bases", "1000", fq], env=dict(os.environ, FLX_CLI_CHUNK_BYTE="1"), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert p.returncode == 1 and b"unknown environment variable FLX_CLI_CHUNK_BYTE" in p.stderr
