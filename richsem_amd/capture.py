"""Stream capture without the cyclic garbage collector in it.

`torch.cuda.graph.__enter__` of this PyTorch (2.10: torch/cuda/graphs.py) only collects garbage before a capture when
`torch.compiler.config.force_cudagraph_gc` is set.  A collection that starts INSIDE the capture -- any Python allocation can
trigger one -- then destroys whatever unreachable cycles earlier work of the process left behind, and if one of them holds a
`torch.cuda.CUDAGraph` (a previous `make_graphed_callables`, with its autograd.Function classes and closures, is exactly such a
cycle) its destructor calls hipGraphExecDestroy on a capturing thread: the HIP error is thrown from a destructor and the process
aborts ("Fatal Python error: Aborted ... Garbage-collecting", seen in 2 of 4 runs of tests/test_gpu_step.py on one box).
"""
import contextlib
import gc


@contextlib.contextmanager
def quiet_gc():
    """collect now, then keep the cyclic collector off for the body (a capture); reference counting still frees what the body drops"""
    was_on = gc.isenabled()
    gc.collect()
    gc.disable()
    try:
        yield
    finally:
        if was_on:
            gc.enable()
