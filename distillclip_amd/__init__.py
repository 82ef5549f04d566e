"""distillclip_amd — MI355X-native distill step of ForJadeForest/DistillCLIP."""
import os as _os

# The four towers (and the RCCL side stream) run on separate HIP streams.  ROCm maps streams onto GPU_MAX_HW_QUEUES hardware
# queues (default 4), and two streams that share a queue serialise: with the default, the two student backwards never
# overlapped and the forward ran 3 instead of 4 kernels wide (rocprofv3 timeline).  The variable is read when
# the HIP runtime initialises, i.e. at the first CUDA call, so importing this package before touching the GPU is enough; an
# explicit setting in the environment wins.
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
