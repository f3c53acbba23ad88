"""TEST INFRASTRUCTURE ONLY (tests/test_asan_cpu.py).  Imported by every Python process started with tools/asan/pyhook on
PYTHONPATH -- the pytest process and the gloo ranks it spawns alike: routes the library's eight host-only index-work entry points
to tools/asan/libsgm_plan_asan.so (the same statements, sgm_plan_host.hpp, under AddressSanitizer + UBSan) and the oracle to
tools/asan/liborc_asan.so.  Active only when SGM_ASAN_HOOK=1; the process must run under LD_PRELOAD=libasan.so."""
import os

if os.environ.get("SGM_ASAN_HOOK") == "1":
    import ctypes

    _here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    PLANNERS = ("sgm_halo_plan_host", "sgm_dist_plan_host", "sgm_dist_neighbors_host", "sgm_partition_links_host",
                "sgm_partition_rows_by_nnz", "sgm_slice_sched_host", "sgm_ell_degrees_host", "sgm_left_permute_rows_host")

    import sigma_amd as _sg
    _orig_lib = _sg.lib
    _plan = ctypes.CDLL(os.path.join(_here, "libsgm_plan_asan.so"))

    def _lib():
        L = _orig_lib()
        if not getattr(L, "_asan_planners", False):
            for nm in PLANNERS:
                setattr(L, nm, getattr(_plan, nm))
            L._asan_planners = True
        return L

    _sg.lib = _lib

    import oracle.oracle as _orc
    _orc.build = lambda force=False: os.path.join(_here, "liborc_asan.so")
    print(f"[asan hook] pid {os.getpid()}: host planners -> libsgm_plan_asan.so, oracle -> liborc_asan.so", flush=True)
