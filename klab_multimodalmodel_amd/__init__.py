"""MI355X-native Swin-V2 -> T5 caption-training hot path (drop-in for the reference's models/model.py).

Runtime configuration note (no longer applied at import: importing a library must not edit the host process's environment).
The engine keeps three HIP streams busy (main, weight-gradient side stream, language-encoder side stream) next to torch's and,
under DDP, ProcessGroupNCCL's stream.  The ROCm runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4);
once streams share a queue the side-stream overlap is lost -- measured: merely creating the RCCL process group cost +0.27 ms per
6.8 ms step with 4 queues and nothing with 8.  The variable is read when the HIP runtime initialises, so the LAUNCHER exports it
(`GPU_MAX_HW_QUEUES=8 torchrun ... train.py`); `bench.py` and `tests/conftest.py` set it themselves before
their first HIP call.  Not under hipGraph replay (KLAB_GRAPH=1): 16.6 ms/step with 8 queues against 6.9 with 4.
"""

__all__ = ["MyModel"]


def __getattr__(name):
    if name == "MyModel":
        from .models.model import MyModel
        return MyModel
    raise AttributeError(name)
