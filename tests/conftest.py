import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
# fixtures of the OPERATOR (value, shapes, lsi, loc, aw, grad_out -> out, grads): every .npz that is not a fixture of another row
_OTHER_ROWS = ("module_", "attnpool_", "clip_resnet_", "layer_", "layer256_", "decoder_stack", "decoder256_", "dn_", "matcher_", "cls_", "frozenbn_", "criterion_")
OP_CASES = sorted(f[:-4] for f in os.listdir(GOLDEN) if f.endswith(".npz") and not f.startswith(_OTHER_ROWS))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # a GPU test collected on a box without a GPU is skipped, never silently passed on a fallback
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _library_options_from_env():
    """RICHSEM_MSDA_OPTS="key=value,key=value": tuning options applied before every test (a whole-suite run of a non-default kernel shape,
    e.g. rps_lanes_per_point=2); the tests' own set_option calls still win where they set the same key"""
    opts = os.environ.get("RICHSEM_MSDA_OPTS", "")
    if opts:
        from richsem_amd import _lib
        for kv in opts.split(","):
            k, v = kv.split("=")
            _lib.set_option(k.strip(), int(v))
    yield
