"""Drop-in alias: `import r3m` / `from r3m import load_r3m` resolve to the MI355X-native implementation (r3m_amd),
so downstream code written against facebookresearch/r3m runs unchanged."""
from r3m_amd import R3M, VALID_ARGS, cleanup_config, load_r3m, load_r3m_reproduce, remove_language_head  # noqa: F401
