import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long CPU test")
    if os.environ.get("PYTEST_XDIST_WORKER"):
        # one worker per core: the oracle's torch CPU ops must not start a thread pool of their own in every worker
        for v in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS"):
            os.environ.setdefault(v, "1")


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """The CPU suite (-m "not gpu") is dominated by single-threaded runs of the kernel emulator: deal its tests to one
    pytest-xdist worker per core unless the caller chose -n himself. A run that selects GPU tests stays in ONE process:
    one process owns the device, and its timing assertions must not share it."""
    if hasattr(config, "workerinput") or not config.pluginmanager.hasplugin("xdist"):
        return None
    if getattr(config.option, "numprocesses", None) is not None or getattr(config.option, "collectonly", False):
        return None
    if os.environ.get("PIPER_TESTS_SERIAL"):
        return None
    expr = (config.option.markexpr or "").replace(" ", "")
    if "notgpu" not in expr:
        return None
    n = min(8, os.cpu_count() or 1)
    if n > 1:
        config.option.numprocesses = n
        if getattr(config.option, "dist", "no") == "no":
            config.option.dist = "load"
    return None


# CPU tests that emulate the most kernel work, longest first (seconds on one core of the build container): started
# first so that the xdist run ends with short tests instead of one long straggler.
_LONGEST_FIRST = [
    "test_cpp_piper_api_on_emulator", "test_emulated_attention_conv_o_layernorm_in_one_launch[lens1",
    "test_warmup_presizes_and_leaves_results_unchanged", "test_emulated_bf16x3_matrix_mode[tiny-high",
    "test_xcd_aware_ffn_slice_order_is_bit_identical", "test_emulated_attention_conv_o_layernorm_in_one_launch[lens0",
    "test_jsonl_drivers_on_emulator", "test_small_call_kernels_do_not_depend_on_wave_order",
    "test_emulated_192_channel_small_call_kernels", "test_engine_group_matches_single_engine",
    "test_emulated_one_tap_convs_without_lds_are_bit_identical", "test_emulated_multi_tile_conv_pipeline",
    "test_emulated_upconv_epilogues_are_bit_identical", "test_workspace_capacities_stay_inside_the_budget",
    "test_emulated_last_res_skip_conv_in_front_of_the_chain",
]


def pytest_collection_modifyitems(config, items):
    def rank(item):
        for i, frag in enumerate(_LONGEST_FIRST):
            if frag in item.nodeid:
                return i
        return len(_LONGEST_FIRST)
    if not any(m.name == "gpu" for it in items for m in it.iter_markers()) or "not gpu" in (config.option.markexpr or ""):
        items.sort(key=rank)         # stable: everything else keeps its file order


@pytest.fixture(scope="session")
def repo_root():
    return ROOT
