"""Drop-in shim: ``from clusterfusion import llama_decoder_layer`` (what the reference's
chat/llama/model.py:19 and tests/test_llama.py:6 import) resolves to the MI355X build.
Mirrors /root/reference/clusterfusion/__init__.py:6-16, which re-exports every public name of
its compiled module."""
from clusterfusion_amd import (  # noqa: F401
    llama_decoder_layer,
    llama_decoder_layer_batch_decode_sglang,
    llama_decoder_layer_sglang,
    rmsnorm,
    deepseek_decoder_layer,
)

__all__ = ["llama_decoder_layer", "llama_decoder_layer_sglang", "llama_decoder_layer_batch_decode_sglang", "rmsnorm",
           "deepseek_decoder_layer"]
