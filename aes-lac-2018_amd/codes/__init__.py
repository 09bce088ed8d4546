"""Host-side mirror of the reference's ``codes`` package for the hot path (MI355X build)."""
