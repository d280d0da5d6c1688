"""Parts of bench.py (harness / single-GPU legs / multi-GPU): measurement code, not the product."""
