"""Input pipeline of the reference (P/misc/dataloader/) for the MI355X path: see dataloader.py."""
