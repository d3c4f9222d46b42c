"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the When2com forward path.

Nothing under ``oracle/`` is product code.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it, and there only as the checker / the timed CPU baseline.  The product
package (``multiagentperception_amd``) never imports from here and raises when
its HIP library is missing.

Parity status: PINNED by outputs of the reference itself.  The reference ships
no tests or golden vectors, so ``oracle/make_golden.py``
imports the reference's own ``ptsemseg.models`` from /root/reference in the
build container (with stub ``torchvision`` / ``pretrainedmodels`` modules and
identity ``.cuda()`` shims), runs it on seeded inputs with the deterministic
weight filler of ``oracle/filler.py`` and commits the resulting small vectors
under ``tests/golden/``.  ``tests/test_oracle_golden.py`` checks this
restatement against those vectors.  The ResNet-18 arithmetic itself lives in
the un-vendored third-party ``pretrainedmodels`` (unpinned,
requirements.txt:11) -> ``torchvision==0.2.0`` ``models.resnet18``
(requirements.txt:5); it is pinned only relative to the standard torchvision
BasicBlock definition restated in ``make_golden.py``.
"""
