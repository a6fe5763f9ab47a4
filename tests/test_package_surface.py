"""CPU: the custom-node package surface ComfyUI reads (reference: __init__.py:24-48): NODE_CLASS_MAPPINGS and, per node class,
INPUT_TYPES / RETURN_TYPES / FUNCTION / CATEGORY with the reference's widget names."""
import cfi_amd


def test_node_class_mappings():
    m = cfi_amd.NODE_CLASS_MAPPINGS
    assert set(m) == {"RIFE VFI", "FILM VFI", "M2M VFI", "IFRNet VFI", "GMFSS Fortuna VFI", "IFUnet VFI", "Make Interpolation State List"}
    assert set(cfi_amd.NODE_DISPLAY_NAME_MAPPINGS) <= set(m)
    for name, cls in m.items():
        it = cls.INPUT_TYPES()
        assert "required" in it and hasattr(cls, "RETURN_TYPES") and isinstance(cls.FUNCTION, str) and hasattr(cls, cls.FUNCTION)
        if name.endswith("VFI"):
            req = list(it["required"])
            assert req[:2] == ["ckpt_name", "frames"] and "multiplier" in req and cls.RETURN_TYPES == ("IMAGE",) and cls.FUNCTION == "vfi"
            assert cls.CATEGORY == "ComfyUI-Frame-Interpolation/VFI" and "optional_interpolation_states" in it["optional"]


def test_widget_lists_match_the_reference():
    m = cfi_amd.NODE_CLASS_MAPPINGS
    req = lambda n: list(m[n].INPUT_TYPES()["required"])   # noqa: E731
    assert req("RIFE VFI") == ["ckpt_name", "frames", "clear_cache_after_n_frames", "multiplier", "fast_mode", "ensemble", "scale_factor",
                               "dtype", "torch_compile", "batch_size"]                                  # rife/__init__.py:36-66
    assert req("IFRNet VFI") == ["ckpt_name", "frames", "clear_cache_after_n_frames", "multiplier", "scale_factor"]      # ifrnet/__init__.py:13-25
    assert req("IFUnet VFI") == ["ckpt_name", "frames", "clear_cache_after_n_frames", "multiplier", "scale_factor", "ensemble"]   # ifunet
    assert req("GMFSS Fortuna VFI") == ["ckpt_name", "frames", "clear_cache_after_n_frames", "multiplier"]           # gmfss_fortuna
    assert m["GMFSS Fortuna VFI"].INPUT_TYPES()["required"]["ckpt_name"][0] == ["GMFSS_fortuna_union", "GMFSS_fortuna"]
    assert req("M2M VFI") == ["ckpt_name", "frames", "clear_cache_after_n_frames", "multiplier"]


def test_rife_launch_size():
    """tasks per launch of the RIFE node: the widget bounded to the library's range, a floor of 8 (4 from 4K up), and at least
    four launches per host clip when it has the tasks for it (upload / compute / download overlap)"""
    from cfi_amd.rife import effective_batch

    assert effective_batch(1, 1080, 1920) == 8 and effective_batch(64, 1080, 1920) == 32 and effective_batch(1, 2160, 4096) == 4
    assert effective_batch(16, 1080, 1920, 32) == 8          # a 33-frame clip: four launches of 8
    assert effective_batch(16, 1080, 1920, 500) == 16        # long clips keep the widget's size
    assert effective_batch(32, 1080, 1920, 100) == 25
    assert effective_batch(16, 1080, 1920, 3) == 8           # (the launch is simply short)


def test_rife_launch_ramp():
    """half-size first and last launch for clips of two launches or more; every task exactly once, none above the configured size"""
    from cfi_amd.rife import launch_sizes

    assert launch_sizes(32, 8) == [4, 8, 8, 8, 4] and launch_sizes(16, 8) == [4, 8, 4] and launch_sizes(17, 8) == [4, 8, 5]
    assert launch_sizes(15, 8) == [8, 7] and launch_sizes(3, 8) == [3] and launch_sizes(7, 2) == [2, 2, 2, 1] and launch_sizes(0, 8) == []
    for n in range(0, 200):
        for bs in (1, 2, 3, 4, 8, 16, 32):
            sz = launch_sizes(n, bs)
            assert sum(sz) == n and all(0 < x <= bs for x in sz)
