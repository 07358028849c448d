"""open_musiclm_amd -- MI355X-native implementation of open-musiclm's TokenConditionedTransformer hot path.

Import as ``open_musiclm_amd`` or, to run the reference's scripts unchanged, as ``open_musiclm`` (the sibling
``open_musiclm/`` package aliases every submodule to this one).
"""
__version__ = "0.1.0"

import os as _os

# The host driver on MI355X boxes only supports dmabuf IPC: RCCL's buffer exchange (and CUDA-tensor sharing between processes) fails
# with `hipIpcGetMemHandle: invalid argument` unless the HSA runtime sees this BEFORE it initialises, i.e. before the first HIP call
# of the process -- so it is set at package import, not next to init_process_group (the model is already on the GPU by then).
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
