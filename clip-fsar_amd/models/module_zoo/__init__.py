"""The reference imports its module zoo here for registry side effects (reference models/module_zoo/__init__.py:4-6).
Nothing of the zoo (TAda/SlowFast/... branches, heads, stems) is on the CLIP-FSAR hot path,
so this package only exists to keep the import surface."""
