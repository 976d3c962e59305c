# Tracks the PROTEUS release whose per-pixel path this drop-in reproduces
# (src/proteus/version.py:1).
VERSION = '1.0.2'
