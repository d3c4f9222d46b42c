"""Import-path shim: lets the reference's callers (`from ptsemseg.models import get_model`,
train.py:16 / test.py:11) resolve to the MI355X implementation.  See INTEGRATION.md."""
