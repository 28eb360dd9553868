"""Minimal stand-in for the `future` package (absent in this image).

Test infrastructure only: lets oracle/make_golden.py import the unmodified
reference from /root/reference. Never imported by sporco_amd.
"""
