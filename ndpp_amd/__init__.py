"""ndpp_amd -- MI355X (gfx950) implementation of NDPP's scattering-moment
integration hot path, behind a C ABI (include/ndpp_hip.h)."""
from .lib import (_check, Params, Stats, NdppError, load, library_path, mu_grid,  # noqa: F401
                  integrate_freegas_leg, integrate_file4_cm_leg,
                  elastic_leg_batch, elastic_leg_batch_device,
                  file6_leg_batch, law9_leg_batch, SabFlat, sab_batch, apply_tol_scatt, ChiSpectrum, ChiNuclide,
                  chi_structs, chi_batch, AceReaction, scattdata_shape, convert_distro,
                  SdGrid, merge_grids, create_ein_grid, AceNuclide, scatt_nuclide, scatt_library,
                  elastic_leg_multi, elastic_leg_multi_device,
                  group_index, scatt_wire, chi_wire, header_wire, DeviceArray, thin_grid, sab_egrid_lib, chi_egrid_lib,
                  OutputOptions, FMT_ASCII, FMT_BINARY, FMT_NONE, scatt_ascii, chi_ascii, header_ascii,
                  real_to_str, ascii_array, lib_xml, finish_scatt, nuclide_file,
                  set_device, freegas_rough_rows, mapped_runtimes, profile_reset, profile_get,
                  ST_NONFINITE, ST_RANGE, ST_ORDER_NOISE)
from .scatt import binary_search, elastic_brackets, calc_elastic_grid  # noqa: F401

__version__ = "0.2.0"
from .grid import merge, add_one_more_point, sab_egrid, chi_egrid  # noqa: F401,E402
