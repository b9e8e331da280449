"""Mirror of the reference's ``thermo_nerf.thermal_nerf`` package (the ThermoNeRF model, field, head, renderer)."""
