import os as _os

# Software-pipelined launches (jb_engine_pipeline) want every stream that carries them on a hardware queue of its own; HIP
# multiplexes streams onto GPU_MAX_HW_QUEUES queues (4 by default), read when the runtime starts.  Raising it here only
# helps when this package is imported before the first HIP call; engines verify their streams either way and keep the
# plain launch chain when they cannot be told apart.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
