"""ORACLE — test infrastructure only (see oracle/README.md).  Never import from the product package."""
