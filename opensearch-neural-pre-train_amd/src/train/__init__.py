"""Training package mirror (V33 DDP trainer only; legacy trainers of the reference are out of scope)."""
