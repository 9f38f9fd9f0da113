"""Source-level guards for defects that tests only catch by timing (no GPU needed)."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = sorted(glob.glob(os.path.join(ROOT, "cuda-sfm_amd", "csrc", "**", "*.*"), recursive=True))


def _code(path):
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return "\n".join(ln.split("//")[0] for ln in text.splitlines())


def test_no_null_stream_memory_calls_in_the_library():
    """hipMemset / hipMemcpy / hipMemcpy2D (the forms without a stream) run on the NULL stream, which streams created with
    hipStreamNonBlocking do not wait for: round 6's second-slot race was one hipMemset (profiles/r06_pipelined_burst_case.txt).  Everything
    the library enqueues names its stream."""
    offenders = []
    for path in SRC:
        if not path.endswith((".hip", ".cpp", ".hpp", ".h")):
            continue
        for m in re.finditer(r"\bhip(Memset|Memcpy|Memcpy2D|MemsetD8|MemsetD32)\s*\(", _code(path)):
            offenders.append(f"{os.path.relpath(path, ROOT)}: {m.group(0)}")
    assert not offenders, offenders


def test_the_product_sources_do_not_reach_for_the_oracle_or_a_cpu_fallback():
    """The product path must fail loudly without the HIP library: nothing under cuda-sfm_amd/ loads, imports or executes anything under oracle/."""
    offenders = []
    for path in sorted(glob.glob(os.path.join(ROOT, "cuda-sfm_amd", "**", "*.*"), recursive=True)):
        if not path.endswith((".hip", ".cpp", ".hpp", ".h", ".py")):
            continue
        code = _code(path) if not path.endswith(".py") else "\n".join(ln.split("#")[0] for ln in open(path).read().splitlines())
        if re.search(r"(import\s+oracle|from\s+oracle|libsfm_oracle|oracle/|sfm_oracle)", code):
            offenders.append(os.path.relpath(path, ROOT))
    assert not offenders, offenders
