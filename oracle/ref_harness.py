"""TEST INFRASTRUCTURE (oracle side) -- import-by-path harness for the upstream reference.

Only usable in the build container, where /root/reference exists.  It never
travels to the GPU box: nothing under tests/ -m gpu, smoke() or bench.py may
import this module.  Used by oracle/gen_golden.py (fixture generation) and by
the `reference`-marked CPU tests that pin the C/numpy oracle to the reference.

The reference's top-level package `gsv_tts/__init__.py` pulls in `av` and
`torchaudio`, which are absent here; its hot-path modules import fine when the
package __init__ is bypassed.
"""
import os
import sys
import types

REF_ROOT = os.environ.get("GSV_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "gsv_tts", "GPT_SoVITS"))


def import_reference():
    """Return (Text2SemanticDecoder, sample, SynthesizerTrn) classes from the reference."""
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REF_ROOT)
    if "gsv_tts" not in sys.modules or not getattr(sys.modules["gsv_tts"], "_gsv_ref_stub", False):
        pkg = types.ModuleType("gsv_tts")
        pkg.__path__ = [os.path.join(REF_ROOT, "gsv_tts")]
        pkg._gsv_ref_stub = True
        sys.modules["gsv_tts"] = pkg
    from gsv_tts.GPT_SoVITS.GPT.t2s_model import Text2SemanticDecoder  # noqa
    from gsv_tts.GPT_SoVITS.GPT.utils import sample  # noqa
    from gsv_tts.GPT_SoVITS.SoVITS.models import SynthesizerTrn  # noqa
    return Text2SemanticDecoder, sample, SynthesizerTrn


def reference_functions(relpath, names, namespace=None):
    """Compile the named function definitions (module-level or methods) of a reference file WITHOUT importing
    the module -- for files whose imports need packages this container lacks (gsv_tts/TTS.py needs `av` and
    `torchaudio`).  The source is read from the reference tree at call time and executed in `namespace`;
    nothing is written anywhere.  Returns {name: function}."""
    import ast
    path = os.path.join(REF_ROOT, relpath)
    with open(path) as f:
        tree = ast.parse(f.read(), filename=path)
    want, found = set(names), []
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name in want:
            found.append(node)
    mod = ast.Module(body=found, type_ignores=[])
    ns = dict(namespace or {})
    exec(compile(mod, path, "exec"), ns)
    missing = want - set(ns)
    if missing:
        raise RuntimeError("not found in %s: %s" % (relpath, sorted(missing)))
    return {n: ns[n] for n in names}
