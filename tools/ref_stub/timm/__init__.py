"""Minimal stand-in for the `timm` package (absent in this image).

Test/tooling infrastructure only: lets scripts under tools/ import the
upstream reference from /root/reference to generate golden vectors.  The
reference uses exactly four helpers from timm; they are restated here.
"""
