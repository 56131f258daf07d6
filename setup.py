#!/usr/bin/env python
"""Packaging metadata of praline_amd.  The part that matters for the drop-in story is the `praline.type` entry-point
group: the reference's TypeIndex.autoregister() registers every entry point of that group
(praline/core/manager.py:72-85, group name praline/core/component.py ENTRY_POINT_GROUP; the reference's own list is
setup.py:8-20), so installing this package next to the reference puts these components - same type ids - into the
reference's index without touching its source.  The HIP library is built in-tree by `make -C praline_amd/csrc`
(see __graft_entry__.build), not by setuptools."""
from setuptools import setup

COMPONENTS = ["PairwiseAligner", "RawPairwiseAligner", "ProfileBuilder", "DummyMasterSlaveAligner",
              "GlobalMasterSlaveAligner", "LocalMasterSlaveAligner", "GuideTreeBuilder",
              "TreeMultipleSequenceAligner", "AdHocMultipleSequenceAligner"]

setup(
    name="praline-amd",
    version="0.2.0",
    description="MI355X-native pairwise profile-profile DP path for PRALINE 2 (HIP, gfx950)",
    packages=["praline_amd"],
    package_data={"praline_amd": ["libpraline_dp.so"]},
    install_requires=["numpy"],
    entry_points={"praline.type": ["{0} = praline_amd.component:{0}".format(name) for name in COMPONENTS]},
)
