"""Host-side mirrors of the reference's misc/ modules on the MI355X path: criteria, optimiser schedules, self-critical reward."""
