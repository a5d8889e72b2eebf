"""CPU: the yaml config front end keeps the reference's API and semantics (pcdet/config.py): _BASE_CONFIG_ include, recursive
section merge with wholesale list replacement, --set overrides with type preservation, attribute access."""
import logging

import pytest

from pcdet.config import AttrDict, cfg_from_list, cfg_from_yaml_file, log_config_to_file


def test_yaml_merge_base_include_and_overrides(tmp_path):
    base = tmp_path / "base.yaml"
    base.write_text("DATA:\n  RANGE: [0, 1, 2]\n  NAME: kitti\n  AUG: {FLIP: true, SCALE: [0.95, 1.05]}\n")
    main = tmp_path / "main.yaml"
    main.write_text(f"_BASE_CONFIG_: {base}\nMODEL:\n  NAME: GDMAE\n  VFE: {{MLPS: [[64, 128]], EPS: 0.001}}\n"
                    "DATA:\n  RANGE: [5, 6]\n  AUG: {FLIP: false}\nOPT: {LR: 0.003, STEPS: [35, 45]}\n")
    cfg = cfg_from_yaml_file(str(main), AttrDict())
    assert cfg.MODEL.NAME == "GDMAE" and cfg.MODEL.VFE.MLPS == [[64, 128]] and isinstance(cfg.MODEL.VFE, AttrDict)
    assert cfg.DATA.NAME == "kitti"                                   # from the base file
    assert cfg.DATA.RANGE == [5, 6]                                   # lists replace wholesale
    assert cfg.DATA.AUG.FLIP is False and cfg.DATA.AUG.SCALE == [0.95, 1.05]      # sections merge
    assert cfg.get("MISSING", None) is None and cfg.MODEL.get("VFE").EPS == 0.001
    cfg_from_list(["OPT.LR", "0.01", "MODEL.NAME", "CenterPoint", "OPT.STEPS", "10,20,30", "DATA.AUG", "FLIP:1"], cfg)
    assert cfg.OPT.LR == 0.01 and cfg.MODEL.NAME == "CenterPoint" and cfg.OPT.STEPS == [10, 20, 30]
    assert cfg.DATA.AUG.FLIP is True and cfg.DATA.AUG.SCALE == [0.95, 1.05]
    with pytest.raises(AssertionError):
        cfg_from_list(["OPT.NOPE", "1"], cfg)                          # unknown key
    with pytest.raises(AssertionError):
        cfg_from_list(["OPT.LR", "abc"], cfg)                          # type change
    with pytest.raises(AssertionError):
        cfg_from_list(["OPT.LR"], cfg)
    lines = []
    log = logging.getLogger("cfgtest")
    log.info = lambda m, *a: lines.append(m % a if a else m)
    log_config_to_file(cfg, logger=log)
    assert "cfg.OPT.LR: 0.01" in lines and "\ncfg.DATA.AUG = edict()" in lines and "cfg.DATA.AUG.FLIP: True" in lines


def test_cfg_from_list_descends_by_position_not_identity():
    """'A.A' (CPython interns one-character strings: both parts are the SAME object) must set config['A']['A']."""
    from pcdet.config import AttrDict, cfg_from_list
    cfg = AttrDict({'A': AttrDict({'A': 1, 'B': 2}), 'B': 3})
    cfg_from_list(['A.A', '5', 'B', '7'], cfg)
    assert cfg.A.A == 5 and cfg.A.B == 2 and cfg.B == 7
