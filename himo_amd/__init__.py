"""himo_amd -- MI355X-native motion-compensation hot path for HiMo."""
import os as _os

# The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), and streams that share a queue
# serialise -- a FALSE dependency between, say, the trainer's critical chain and one of its three weight-gradient streams, or between a
# feeder's copy stream and a pipeline stream.  This package runs up to eight streams at a time (three batches in flight + feeder +
# drain; the trainer's chain + three side streams beside whatever the caller keeps): measured, a training step called from a created
# stream is 8 % slower on four queues than on eight (profiles/r05_exp_stream_priority.txt), and the training leg of the default bench
# line -- which runs in a process that has created the pipeline's streams before -- 114 vs 122 frames/s.  Only a default: set before the
# runtime initialises (i.e. import this package before the first device call), never overrides the caller's own setting.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
