"""Inference-side mirror of ref:benchmark/ — only the V33 sparse encoder; the OpenSearch
benchmark harness around it is out of scope."""
