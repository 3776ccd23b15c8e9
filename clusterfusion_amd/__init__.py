"""clusterfusion_amd -- MI355X-native (gfx950) implementation of ClusterFusion's fused Llama
decoder-layer decode op, behind the reference's own operator API.

    from clusterfusion_amd import llama_decoder_layer            # == reference clusterfusion.*

Importing the package is cheap and works without a GPU; the first call into an op loads
libclusterfusion_hip.so and fails loudly if it is not built (no CPU / eager fallback).
"""
from .ops import (  # noqa: F401
    algorithmic_bytes,
    PreparedLayer,
    decoder_layer,
    prepare_decoder_layer,
    llama_decoder_layer,
    llama_decoder_layer_batch_decode_sglang,
    llama_decoder_layer_sglang,
    rmsnorm,
    deepseek_decoder_layer,
    deepseek_algorithmic_bytes,
    deepseek_profile,
    profile_enable,
    profile_read,
    check_device_errors,
    host_binding,
    set_host_binding,
    host_binding_stats,
    last_path,
    last_variant,
    last_arm,
    set_path,
    set_tuning,
    set_weight_relayout,
    invalidate_weight_relayout,
    release_weight_relayout,
    weight_relayout_stats,
    workspace_bytes,
)

__version__ = "0.1.0"
