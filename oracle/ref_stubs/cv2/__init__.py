"""TEST INFRASTRUCTURE ONLY — empty stub (utils.py:31 imports the module object only)."""
