"""open_musiclm_amd -- MI355X-native implementation of open-musiclm's TokenConditionedTransformer hot path.

Import as ``open_musiclm_amd`` or, to run the reference's scripts unchanged, as ``open_musiclm`` (the sibling
``open_musiclm/`` package aliases every submodule to this one).
"""
__version__ = "0.1.0"
