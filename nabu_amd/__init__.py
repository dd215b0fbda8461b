"""nabu_amd — MI355X-native implementation of the Nabu (vrenkens/nabu) training
hot path: Listener/DBLSTM encoders, Speller and DNN decoders, CTC and
cross-entropy losses, clip+Adam, data-parallel RCCL all-reduce — behind the
reference's recipe API (Model / Trainer / EDEncoder / EDDecoder / factories and
.cfg defaults).  All arithmetic runs in hand-written gfx950 HIP kernels exposed
through the C ABI in include/nabu_hip.h."""
__version__ = '0.1.0'
