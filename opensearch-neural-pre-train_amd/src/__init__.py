"""Mirror of the reference's ``src`` package for the SPLADE-ModernBERT training path only.

Same import paths, class names and signatures as the reference (so its trainer drops in), with
the hot path running on hand-written HIP kernels through ``snx`` (the C-ABI library binding).
Unlike the reference's ``src/__init__.py`` nothing is imported eagerly here (the reference's
eager import chain pulls in ``sentence_transformers``)."""
