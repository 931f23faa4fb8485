import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _usable_cores():
    """The affinity mask capped by the cgroup CPU quota (os.cpu_count() reports the host's cores)."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    try:
        txt = open("/sys/fs/cgroup/cpu.max").read().split()
        if txt[0] != "max":
            n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
    except (OSError, ValueError, IndexError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return n


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long CPU test")
    if os.environ.get("PYTEST_XDIST_WORKER"):
        # one worker per core: the oracle's torch CPU ops must not start a thread pool of their own in every worker
        for v in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS"):
            os.environ.setdefault(v, "1")
    elif not getattr(config.option, "numprocesses", None):      # (a controller of xdist workers sets nothing: they would inherit it)
        # a single process (the GPU suite): the oracle's torch CPU ops get the cores this process may really use. On a GPU box
        # whose cgroup grants 16 of 256 cores the default pool (one thread per visible core) made every oracle utterance take
        # 1-2.5 s: four tests were 550 of the suite's 960 s (profiles/r05_notes.md, call 38)
        n = _usable_cores()
        for v in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS"):
            os.environ.setdefault(v, str(n))
        if "torch" in sys.modules:
            try:
                sys.modules["torch"].set_num_threads(n)
            except Exception:
                pass


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
    "test_warmup_presizes_and_leaves_results_unchanged", "test_emulated_split_matrix_modes[tiny-high",
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
