"""placeholder; filled in below"""
