"""multimesh_amd -- MI355X-native drop-in for the MultiMesh mesh-to-mesh interpolation hot path.

``multimesh_amd.api`` mirrors the reference's entry points, ``multimesh_amd.helpers.load_lib`` its
library loader, ``multimesh_amd.device`` wraps the device-pointer C ABI, and
``multimesh_amd.distributed`` shards targets over the GPUs of one node.
"""
__all__ = ["api", "device", "distributed", "helpers", "mesh", "synth"]
