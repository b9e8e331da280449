"""Mirror of the reference's ``thermo_nerf.nerfacto_config`` package (model base class + config surface)."""
