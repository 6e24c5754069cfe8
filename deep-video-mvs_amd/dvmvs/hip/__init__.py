"""Binding layer between the Python surface and ``libdvmvs_hip.so`` (C ABI in ``include/dvmvs_hip.h``)."""
