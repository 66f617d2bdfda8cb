"""MI355X-native retrieval hot path (drop-in for the reference's retrieval components).

The importable package name uses underscores; the hyphenated spelling `rag-inference-pipeline_amd`
is not a valid Python identifier.
"""

__version__ = "0.1.0"
