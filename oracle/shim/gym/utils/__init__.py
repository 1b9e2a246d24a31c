from . import seeding
