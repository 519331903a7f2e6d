"""Shapes of the absorption-coefficient ("k-data") arrays of RRTMG_LW, original 16-g-point form.

One entry per array that the 16 ``lw_kgbNN`` routines fill (reference: modules/rrlw_kg01.f90:31-35 ...
modules/rrlw_kg16.f90:28-34; netCDF reader src/rrtmg_lw_read_nc.f90:47-1059).  In the k-data blob
(``*.kdata.bin``) every array is stored under ``bNN.<name>`` with exactly these Fortran bounds.

kind 'k': reduced 256->140 with the rwgt weights (src/rrtmg_lw_init.f90:385-2034, e.g. :467-478)
kind 'f': Planck fractions, reduced by plain summation   (e.g. src/rrtmg_lw_init.f90:693-712)
``gdim`` is the position of the g-point axis (0 = first/fastest, -1 = last).
"""

NG_ORIG = 16
NGC = (10, 12, 16, 14, 16, 8, 12, 8, 12, 6, 8, 8, 4, 2, 2, 2)   # src/rrtmg_lw_init.f90:325

_KA1 = [(1, 5), (1, 13), (1, 16)]
_KB1 = [(1, 5), (13, 59), (1, 16)]
_KA9 = [(1, 9), (1, 5), (1, 13), (1, 16)]
_KB5 = [(1, 5), (1, 5), (13, 59), (1, 16)]
_M1 = [(1, 19), (1, 16)]
_M9 = [(1, 9), (1, 19), (1, 16)]
_M5 = [(1, 5), (1, 19), (1, 16)]
_SELF = [(1, 10), (1, 16)]
_FOR = [(1, 4), (1, 16)]
_V = [(1, 16)]
_F9 = [(1, 16), (1, 9)]
_F5 = [(1, 16), (1, 5)]


def _k(name, bounds):
    return (name, bounds, "k", -1)


def _f(name, bounds):
    return (name, bounds, "f", 0)


_COMMON = [_k("selfrefo", _SELF), _k("forrefo", _FOR)]

KSPEC = {
    1: [_f("fracrefao", _V), _f("fracrefbo", _V), _k("kao", _KA1), _k("kbo", _KB1),
        _k("kao_mn2", _M1), _k("kbo_mn2", _M1)] + _COMMON,
    2: [_f("fracrefao", _V), _f("fracrefbo", _V), _k("kao", _KA1), _k("kbo", _KB1)] + _COMMON,
    3: [_f("fracrefao", _F9), _f("fracrefbo", _F5), _k("kao", _KA9), _k("kbo", _KB5),
        _k("kao_mn2o", _M9), _k("kbo_mn2o", _M5)] + _COMMON,
    4: [_f("fracrefao", _F9), _f("fracrefbo", _F5), _k("kao", _KA9), _k("kbo", _KB5)] + _COMMON,
    5: [_f("fracrefao", _F9), _f("fracrefbo", _F5), _k("kao", _KA9), _k("kbo", _KB5),
        _k("kao_mo3", _M9), _k("ccl4o", _V)] + _COMMON,
    6: [_f("fracrefao", _V), _k("kao", _KA1), _k("kao_mco2", _M1), _k("cfc11adjo", _V),
        _k("cfc12o", _V)] + _COMMON,
    7: [_f("fracrefao", _F9), _f("fracrefbo", _V), _k("kao", _KA9), _k("kbo", _KB1),
        _k("kao_mco2", _M9), _k("kbo_mco2", _M1)] + _COMMON,
    8: [_f("fracrefao", _V), _f("fracrefbo", _V), _k("cfc12o", _V), _k("cfc22adjo", _V),
        _k("kao", _KA1), _k("kao_mco2", _M1), _k("kao_mn2o", _M1), _k("kao_mo3", _M1),
        _k("kbo", _KB1), _k("kbo_mco2", _M1), _k("kbo_mn2o", _M1)] + _COMMON,
    9: [_f("fracrefao", _F9), _f("fracrefbo", _V), _k("kao", _KA9), _k("kbo", _KB1),
        _k("kao_mn2o", _M9), _k("kbo_mn2o", _M1)] + _COMMON,
    10: [_f("fracrefao", _V), _f("fracrefbo", _V), _k("kao", _KA1), _k("kbo", _KB1)] + _COMMON,
    11: [_f("fracrefao", _V), _f("fracrefbo", _V), _k("kao", _KA1), _k("kbo", _KB1),
         _k("kao_mo2", _M1), _k("kbo_mo2", _M1)] + _COMMON,
    12: [_f("fracrefao", _F9), _k("kao", _KA9)] + _COMMON,
    13: [_f("fracrefao", _F9), _f("fracrefbo", _V), _k("kao", _KA9), _k("kao_mco2", _M9),
         _k("kao_mco", _M9), _k("kbo_mo3", _M1)] + _COMMON,
    14: [_f("fracrefao", _V), _f("fracrefbo", _V), _k("kao", _KA1), _k("kbo", _KB1)] + _COMMON,
    15: [_f("fracrefao", _F9), _k("kao", _KA9), _k("kao_mn2", _M9)] + _COMMON,
    16: [_f("fracrefao", _F9), _f("fracrefbo", _V), _k("kao", _KA9), _k("kbo", _KB1)] + _COMMON,
}


def blob_name(band, name):
    return f"b{band:02d}.{name}"


def shape_of(bounds):
    return tuple(hi - lo + 1 for lo, hi in bounds)


def all_entries():
    for band in range(1, 17):
        for name, bounds, kind, gdim in KSPEC[band]:
            yield band, name, bounds, kind, gdim
